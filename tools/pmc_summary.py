"""Summarise the rocprofv3 PMC passes of tools/pmc_traffic.sh into profiles/<round>_traffic.json (bench.py reads it).

    python tools/pmc_summary.py cfg2b 4096 [round tag, default r2]

FETCH_SIZE / WRITE_SIZE are reported in KB per dispatch; on gfx950 FETCH_SIZE counts half of the bytes of a coalesced
streaming read (MI355X_MICROARCH.md, HBM section), so it is doubled - an ASSUMPTION for this kernel's 4-byte-per-lane
sweeps that the byte model of DESIGN.md section 6 supports but does not prove; WRITE_SIZE is taken as is."""
import collections
import csv
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, n_img = sys.argv[1], int(sys.argv[2])
rnd = sys.argv[3] if len(sys.argv) > 3 else "r3"
base = os.path.join(REPO, "gpurun_out", f"pmc_{tag}")
acc = collections.defaultdict(list)
for name in ("fetch", "write", "sq"):
    found = [os.path.join(d, f) for d, _, fs in os.walk(os.path.join(base, name)) for f in fs if f.endswith("counter_collection.csv")]
    if not found:
        continue
    path = found[0]
    for r in csv.DictReader(open(path)):
        if "k_raster_dense<2>" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            acc["_vgpr"].append(float(r["VGPR_Count"])); acc["_lds"].append(float(r["LDS_Block_Size"])); acc["_grid"].append(float(r["Grid_Size"]))
mean = {k: sum(v) / len(v) for k, v in acc.items()}
missing = [c for c in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU") if c not in mean]
if missing:  # a partial set of passes must not become the bench line's `traffic`
    raise SystemExit(f"pmc_summary: no rows for {missing} under {base}: all three passes of tools/pmc_traffic.sh are required")
out_path = os.path.join(REPO, "profiles", f"{rnd}_traffic.json")
data = json.load(open(out_path)) if os.path.exists(out_path) else {}
entry = {"source": f"rocprofv3 --pmc, separate passes (tools/pmc_traffic.sh {tag}); kernel k_raster_dense<2>, mean over {len(acc.get('FETCH_SIZE', []))} dispatches",
         "images_per_launch": n_img, "FETCH_SIZE_KB": mean.get("FETCH_SIZE"), "WRITE_SIZE_KB": mean.get("WRITE_SIZE"), "fetch_correction": 2.0,
         "fetch_correction_note": "calibrated for 4 B/lane coalesced reads: profiles/r2_fetch_calib.txt (tools/dbg/fetch_calib.hip)",
         "sq": {k: v for k, v in mean.items() if k.startswith("SQ_")}, "vgpr": mean.get("_vgpr"), "lds_bytes": mean.get("_lds"), "grid": mean.get("_grid")}
data[tag] = entry
os.makedirs(os.path.dirname(out_path), exist_ok=True)
json.dump(data, open(out_path, "w"), indent=1)
print(json.dumps(entry, indent=1))
if entry["FETCH_SIZE_KB"] and entry["WRITE_SIZE_KB"]:
    tot = (entry["FETCH_SIZE_KB"] * 2 + entry["WRITE_SIZE_KB"]) * 1024
    print(f"traffic per launch {tot/1e9:.2f} GB = {tot/n_img/1e3:.1f} KB per image")
