#!/usr/bin/env python3
"""Summarise the rocprofv3 CSVs written by tools/pmc_probe.sh: per-kernel average duration and counter means."""
import csv
import re
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
lines = []


def short(n):
    if "mgx_obs_kernel" in n:
        return "mgx_fused<gen_obs>"
    m = re.search(r"mgx_fused_kernel<\d+, (\d)", n) or re.search(r"mgx_fused_kernelILi\d+ELi(\d)E", n)
    if m:
        return {"0": "mgx_fused<gen_obs>", "1": "mgx_fused<step>", "2": "mgx_fused<rollout>"}[m.group(1)]
    return n[:40]


for f in sorted(glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        if "mgx" in r.get("Name", ""):
            lines.append(f"kernel_stats {short(r['Name'])}: calls={r['Calls']} avg_ns={r['AverageNs']} "
                         f"min_ns={r['MinNs']} max_ns={r['MaxNs']}")
for d in sorted(glob.glob(os.path.join(out, "pmc*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = defaultdict(lambda: defaultdict(list))
        for r in csv.DictReader(open(f)):
            if "mgx" in r["Kernel_Name"]:
                acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            for c, v in sorted(cs.items()):
                lines.append(f"{os.path.basename(d)} {k} {c}: mean={sum(v) / len(v):.6g} n={len(v)}")
txt = "\n".join(lines)
print(txt)
open(os.path.join(out, "summary.txt"), "w").write(txt + "\n")
