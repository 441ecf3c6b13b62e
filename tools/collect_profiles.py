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

# Register / LDS / spill figures of every rasteriser and LBS kernel, straight from the compiler's resource remarks on the shipped sources
# (DESIGN.md section 4.1 quotes this file instead of hand-copied numbers)
import re

hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
lines = []
for unit in ("raster.hip", "lbs.hip", "fit.hip", "project.hip"):
    r = subprocess.run([hipcc, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I" + os.path.join(REPO, "include"), "--cuda-device-only",
                        "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(REPO, "smilify_amd", "csrc", unit), "-o", os.devnull],
                       capture_output=True, text=True)
    cur = {}
    for ln in r.stderr.splitlines():
        m = re.search(r"remark:\s+(Function Name|VGPRs|AGPRs|SGPRs Spill|VGPRs Spill|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", ln)
        if not m:
            continue
        if m.group(1) == "Function Name":
            if cur:
                lines.append(cur)
            cur = {"unit": unit, "kernel": subprocess.run(["c++filt", m.group(2)], capture_output=True, text=True).stdout.strip() or m.group(2)}
        else:
            cur[m.group(1)] = m.group(2)
    if cur:
        lines.append(cur)
if lines:
    with open(os.path.join(dst, f"{tag}_kernel_resources.txt"), "w") as fh:
        fh.write("hipcc -O3 --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage on smilify_amd/csrc (static LDS only; dynamic LDS is set at launch)\n")
        fh.write(f"{'kernel':70s} VGPRs AGPRs vgpr_spill sgpr_spill scratch_B/lane waves/SIMD static_LDS_B\n")
        for c in lines:
            fh.write(f"{c['kernel'][:70]:70s} {c.get('VGPRs', '?'):>5s} {c.get('AGPRs', '?'):>5s} {c.get('VGPRs Spill', '?'):>10s} {c.get('SGPRs Spill', '?'):>10s} "
                     f"{c.get('ScratchSize [bytes/lane]', '?'):>14s} {c.get('Occupancy [waves/SIMD]', '?'):>10s} {c.get('LDS Size [bytes/block]', '?'):>12s}\n")
print(sorted(os.path.basename(p) for p in glob.glob(os.path.join(dst, f"{tag}_*"))))
for w in ("cfg2b", "cfg2", "cfg3", "cfg4", "cfg5s"):
    p = os.path.join(dst, f"{tag}_bench_{w}.json")
    if os.path.exists(p):
        d = json.load(open(p))
        print(w, f"{d['ms_per_step']:.2f} ms/step  {d['value']:.2f} fit-iters/s ({d.get('frame_iters_per_sec', 0):.0f} frame-iters/s)  kernel {d['roofline']['kernel_ms']:.2f} ms  frac {d['roofline']['frac']:.4f}")
