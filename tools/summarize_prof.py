#!/usr/bin/env python3
"""Condense a tools/prof.sh output directory into the small files committed under profiles/."""
import collections, csv, glob, json, os, sys

src, dst, tag = sys.argv[1], sys.argv[2], sys.argv[3]
os.makedirs(dst, exist_ok=True)
stats = glob.glob(os.path.join(src, "trace", "*", "*kernel_stats.csv"))[0]
rows = list(csv.DictReader(open(stats)))
with open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
    for r in rows:
        name = r["Name"]
        short = next((k for k in ("k_trace", "k_project", "k_bin", "k_super_bbox") if k in name), name[:70])
        w.writerow([short, r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"], r["StdDev"]])
import importlib.util
_spec = importlib.util.spec_from_file_location("rtgs_build", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                                           "rt-gaussian-splat-renderer_amd", "build.py"))
_build = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_build)
out = {"source": "rocprofv3 --kernel-trace --stats / --pmc (separate passes), python3 bench.py --no-cpu-baseline --no-secondary",
       # hash of csrc/* + include/rtgs_hip.h the profiled library was built from: bench.py only uses these counters
       # when it is timing a build of the same sources
       "source_hash": _build.source_hash(),
       "kernels": {}}
for r in rows:
    for k in ("k_trace", "k_project", "k_bin", "k_super_bbox"):
        if k in r["Name"]:
            out["kernels"][k] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "min_ns": float(r["MinNs"]),
                                 "max_ns": float(r["MaxNs"])}
pmc = collections.defaultdict(lambda: collections.defaultdict(list))
meta = {}
for f in glob.glob(os.path.join(src, "pmc_*", "*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        for k in ("k_trace", "k_project"):
            if k in r["Kernel_Name"]:
                pmc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
                meta[k] = {x: r[x] for x in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "SGPR_Count", "Scratch_Size") if x in r}
for k, d in pmc.items():
    out["kernels"].setdefault(k, {})["pmc_mean_per_launch"] = {c: sum(v) / len(v) for c, v in sorted(d.items())}
    out["kernels"][k]["dispatch"] = meta.get(k)
t = out["kernels"].get("k_trace", {}).get("pmc_mean_per_launch", {})
if "FETCH_SIZE" in t and "WRITE_SIZE" in t:
    # FETCH_SIZE / WRITE_SIZE are in KiB.  MI355X_MICROARCH.md (HBM): FETCH_SIZE under-reports wide
    # (16 B/lane) coalesced reads by exactly 2x; narrower / gathered accesses are uncalibrated, so
    # both the raw and the doubled figure are given.  WRITE_SIZE is exact for 16-B stores.
    out["k_trace_hbm_bytes_per_launch"] = {
        "fetch_raw": t["FETCH_SIZE"] * 1024, "fetch_x2": 2 * t["FETCH_SIZE"] * 1024,
        "write": t["WRITE_SIZE"] * 1024,
        "traffic_upper": (2 * t["FETCH_SIZE"] + t["WRITE_SIZE"]) * 1024}
json.dump(out, open(os.path.join(dst, f"{tag}_summary.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
