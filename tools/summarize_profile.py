#!/usr/bin/env python3
"""Summarise gpurun_out/<tag>/ (tools/profile_gpu.sh) into profiles/<name>_{kernel_stats.csv,pmc_summary.json}."""
import csv
import glob
import json
import os
import sys

tag, name = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", tag)
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)
KERNEL = "orr_step_kernel"
def newest(pattern):
    """gpurun merges every call's files into the same directory: keep the newest file of each directory"""
    best = {}
    for f in glob.glob(pattern, recursive=True):
        d = os.path.dirname(f)
        if d not in best or os.path.getmtime(f) > os.path.getmtime(best[d]):
            best[d] = f
    return sorted(best.values())


for f in newest(os.path.join(src, "trace", "**", "*kernel_stats.csv")):
    rows = list(csv.reader(open(f)))
    with open(os.path.join(dst, name + "_kernel_stats.csv"), "w") as o:
        w = csv.writer(o)
        w.writerow(rows[0])
        for r in rows[1:6]:
            r[0] = r[0][:120]
            w.writerow(r)
summary = {}
for f in newest(os.path.join(src, "pmc_*", "**", "*counter_collection.csv")):
    rd = csv.DictReader(open(f))
    acc = {}
    for r in rd:
        if KERNEL not in r.get("Kernel_Name", ""):
            continue
        c = r["Counter_Name"]
        acc.setdefault(c, []).append(float(r["Counter_Value"]))
        if "ILi0E" in r.get("Kernel_Name", "") or "<0" in r.get("Kernel_Name", ""):
            for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Grid_Size", "Workgroup_Size"):
                if k in r:
                    summary.setdefault("dispatch", {})[k] = r[k]
    for c, v in acc.items():
        summary[c] = {"mean_per_launch": sum(v) / len(v), "launches": len(v)}
if "FETCH_SIZE" in summary and "WRITE_SIZE" in summary:
    # rocprofv3 reports KiB; gfx950 FETCH_SIZE under-counts wide coalesced reads by 2x (MI355X_MICROARCH.md, HBM):
    # this kernel's reads are 4-byte-per-lane coalesced rows, for which the factor is uncalibrated -> report both
    fe, wr = summary["FETCH_SIZE"]["mean_per_launch"] * 1024, summary["WRITE_SIZE"]["mean_per_launch"] * 1024
    summary["hbm_bytes_per_launch"] = fe + wr
    summary["hbm_bytes_per_launch_fetch_x2"] = 2 * fe + wr
# which kernel sources the counters belong to: tools/profile_gpu.sh records the hash of the library the profiled command loaded;
# bench.py refuses a summary whose hash differs from its own library's ("pmc_stale")
try:
    summary["source_hash"] = open(os.path.join(src, "source_hash.txt")).read().strip()
except OSError:
    summary["source_hash"] = None
json.dump(summary, open(os.path.join(dst, name + "_pmc_summary.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(summary, indent=1, sort_keys=True)[:3000])
