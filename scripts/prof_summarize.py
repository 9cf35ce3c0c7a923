"""Condense rocprofv3 CSV output into small text summaries (kept under profiles/).

usage: prof_summarize.py <rocprof_out_dir> <summary.txt>
Finds *kernel_stats.csv (from --stats) and *counter_collection.csv (from --pmc) below the
directory and writes: per-kernel calls / total / average duration (top 25), and per-kernel
mean counter values for our hand-written kernels (names containing 'tia::').
"""
import csv
import glob
import os
import sys
from collections import defaultdict

src, dst = sys.argv[1], sys.argv[2]
lines = []
for f in sorted(glob.glob(os.path.join(src, "**", "*kernel_stats.csv"), recursive=True)):
    rows = list(csv.DictReader(open(f)))
    lines.append(f"# {os.path.relpath(f, src)}  (rocprofv3 --kernel-trace --stats)")
    lines.append(f"{'calls':>8} {'total_ms':>12} {'avg_us':>12} {'pct':>7}  name")
    for r in rows[:25]:
        name = r.get("Name", "")[:140]
        lines.append(f"{int(r['Calls']):8d} {float(r['TotalDurationNs'])/1e6:12.3f} "
                     f"{float(r['AverageNs'])/1e3:12.2f} {float(r['Percentage']):7.2f}  {name}")
    lines.append("")
for f in sorted(glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True)):
    acc = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    lines.append(f"# {os.path.relpath(f, src)}  (rocprofv3 --pmc), mean per dispatch")
    for k, ctrs in acc.items():
        if "tia" not in k and "conv_mfma" not in k and "stem7x7" not in k and "wino" not in k:
            continue
        for c, vals in ctrs.items():
            lines.append(f"{c:>14} mean={sum(vals)/len(vals):16.1f} n={len(vals):5d}  {k[:120]}")
    lines.append("")
open(dst, "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:60]))
