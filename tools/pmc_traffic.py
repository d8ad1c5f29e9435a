#!/usr/bin/env python3
"""Turn rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into the per-launch HBM traffic of the GEMM kernel.

    python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> [kernel-substring]

FETCH_SIZE / WRITE_SIZE are reported in KiB.  On gfx950 FETCH_SIZE counts 64 B per 128-B request
of a wide coalesced stream, i.e. HALF the bytes (MI355X_MICROARCH.md, section HBM): the read side
is doubled here, WRITE_SIZE is taken as is, as that guide prescribes.
"""
import csv
import json
import sys


def per_launch(path, counter, kernel):
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path))
            if r["Counter_Name"] == counter and kernel in r["Kernel_Name"]]
    return (sum(vals) / len(vals) * 1024.0, len(vals)) if vals else (None, 0)


def main():
    fetch_csv, write_csv, out = sys.argv[1:4]
    kernel = sys.argv[4] if len(sys.argv) > 4 else "gemm_pp_kernel"
    fetch, n = per_launch(fetch_csv, "FETCH_SIZE", kernel)
    write, _ = per_launch(write_csv, "WRITE_SIZE", kernel)
    res = {"kernel": kernel, "launches": n, "fetch_bytes_per_launch_corrected_x2": 2 * fetch, "write_bytes_per_launch": write,
           "hbm_bytes_per_launch": 2 * fetch + write,
           "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes), averaged over the kernel's launches; "
                     "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 64 B per 128-B request)"}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
