#!/usr/bin/env python3
"""Raw launch list of a time window of a rocprofv3 kernel trace (csv): start / end relative to the window's first launch, queue,
kernel.  usage: kernel_window.py <kernel_trace.csv> [--after-gram N] [--us W]: the window starts at the N-th bs_gram_kernel."""
import argparse
import csv


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--after-gram", type=int, default=200)
    ap.add_argument("--us", type=float, default=700.0)
    args = ap.parse_args()
    rows = []
    with open(args.trace) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"]))
    rows.sort()
    grams = [r for r in rows if "bs_gram" in r[3]]
    t0 = grams[min(args.after_gram, len(grams) - 1)][0]
    for s, e, q, name in rows:
        if t0 - 50e3 <= s <= t0 + args.us * 1e3:
            short = name.replace("cu2rec::(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
            print("%9.2f %9.2f  dur %8.2f  q%-3s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, short))


if __name__ == "__main__":
    main()
