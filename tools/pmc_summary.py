#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs (one counter set per pass) into one JSON.

usage: tools/pmc_summary.py OUT.json  DIR_OR_CSV [DIR_OR_CSV ...]

HBM traffic is priced as MI355X_MICROARCH.md (HBM section) prescribes: FETCH_SIZE and WRITE_SIZE are
in KiB, collected in separate passes (they do not fit one pass); on gfx950 FETCH_SIZE counts 128-B
requests at 64 B, i.e. reads exactly half the bytes of a wide (16 B/lane) coalesced read, so it is
doubled; WRITE_SIZE is exact for 16-B-per-lane stores.  Both the raw and the corrected figure are kept.
"""
import collections
import csv
import glob
import json
import os
import sys


def main():
    out_path, inputs = sys.argv[1], sys.argv[2:]
    files = []
    for p in inputs:
        files += [p] if p.endswith(".csv") else glob.glob(os.path.join(p, "**", "*counter_collection.csv"), recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in files:
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    summary = {}
    for kernel, counters in agg.items():
        if "cu2rec" not in kernel:
            continue
        k = {c: {"launches": len(v), "mean": sum(v) / len(v), "min": min(v), "max": max(v)} for c, v in counters.items()}
        if "FETCH_SIZE" in k and "WRITE_SIZE" in k:
            fetch, write = k["FETCH_SIZE"]["mean"] * 1024.0, k["WRITE_SIZE"]["mean"] * 1024.0
            k["hbm_bytes_per_launch_raw"] = fetch + write
            k["hbm_bytes_per_launch_corrected"] = 2.0 * fetch + write
        if "TCC_HIT_sum" in k and "TCC_MISS_sum" in k:
            h, m = k["TCC_HIT_sum"]["mean"], k["TCC_MISS_sum"]["mean"]
            k["l2_hit_rate"] = h / (h + m)
        summary[kernel] = k
    with open(out_path, "w") as fh:
        json.dump(summary, fh, indent=1)
    print(json.dumps({k: {kk: vv for kk, vv in v.items() if not isinstance(vv, dict)} for k, v in summary.items()}, indent=1))


if __name__ == "__main__":
    main()
