#!/usr/bin/env python3
"""L2 hit rate and fabric-side request mix per kernel from one `rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum
TCC_EA0_RDREQ_DRAM_sum` pass.   python tools/pmc_l2.py <counter_collection.csv> <out.json>

What the counters can and cannot say on gfx950 / ROCm 7.2: TCC_EA0_RDREQ counts the L2's read requests to the fabric,
..._DRAM the share routed to the memory controllers (as opposed to GMI / IO).  The Infinity Cache (MALL) sits BEHIND that
routing, on the memory side: there is no per-kernel counter in `rocprofv3 -L` that separates a MALL hit from an HBM access
(gpurun_out/r03_counters.txt lists the TCC counters this stack has).  So this pass bounds the question from the L2 side: the hit
rate says how much of the operand re-streaming the L2 absorbs, RDREQ x request size what goes past it, and the rate at which it
goes past it (bytes / kernel time) says whether HBM alone could have served it."""
import csv
import json
import sys
from collections import defaultdict

src, out = sys.argv[1:3]
acc = defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(src)):
    k = r["Kernel_Name"]
    k = k[:k.index("(")] if "(" in k else k
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, c in acc.items():
    if "TCC_HIT_sum" not in c:
        continue
    n = len(c["TCC_HIT_sum"])
    m = {name: sum(v) / len(v) for name, v in c.items()}
    hit, miss = m.get("TCC_HIT_sum", 0.0), m.get("TCC_MISS_sum", 0.0)
    res[k] = {"launches": n, "l2_hit_rate": round(hit / max(hit + miss, 1.0), 4), "tcc_hit": hit, "tcc_miss": miss,
              "ea_rdreq": m.get("TCC_EA0_RDREQ_sum"), "ea_rdreq_dram": m.get("TCC_EA0_RDREQ_DRAM_sum"),
              "ea_rdreq_dram_share": round(m.get("TCC_EA0_RDREQ_DRAM_sum", 0.0) / max(m.get("TCC_EA0_RDREQ_sum", 1.0), 1.0), 4)}
json.dump(res, open(out, "w"), indent=1)
for k, v in sorted(res.items(), key=lambda kv: -kv[1]["launches"])[:12]:
    print(f"{k[:70]:<70s} n {v['launches']:5d}  L2 hit {v['l2_hit_rate']:.3f}  EA rdreq {v['ea_rdreq']:.3e} (DRAM-routed {v['ea_rdreq_dram_share']:.3f})")
