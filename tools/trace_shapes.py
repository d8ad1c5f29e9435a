#!/usr/bin/env python3
"""Per-(kernel, grid) average durations from a rocprofv3 kernel trace CSV: tools/trace_shapes.py <kernel_trace.csv>"""
import collections
import csv
import sys

d = collections.defaultdict(list)
with open(sys.argv[1]) as f:
    for row in csv.DictReader(f):
        n = row["Kernel_Name"]
        if "gemm" in n or "attn" in n or "step_embed" in n:
            d[(n[:48], row.get("Grid_Size_X") or row.get("Grid_Size"))].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in d.values())
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:10]:
    print(f"{k[0]:48s} grid {k[1]:>8s}  n {len(v):6d}  avg {sum(v) / len(v):7.1f} us  {100 * sum(v) / tot:5.1f} %")
