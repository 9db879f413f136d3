#!/usr/bin/env python3
"""How much do the kernels that build the NEXT batch's schedule (keys, radix sort, plan, chain ranges: their own stream) cost the
block-solve iterations they run beside?  From a rocprofv3 kernel trace: the period of every iteration (start of bs_gram_kernel to
the next one's), split by whether a schedule kernel ran inside it.
usage: schedule_interference.py <kernel_trace.csv> [--skip N]"""
import argparse
import csv

import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--skip", type=int, default=100)
    args = ap.parse_args()
    gram, sched = [], []
    with open(args.trace) as fh:
        for row in csv.DictReader(fh):
            name = row["Kernel_Name"]
            s, e = int(row["Start_Timestamp"]), int(row["End_Timestamp"])
            if "bs_gram" in name:
                gram.append(s)
            elif any(k in name for k in ("schedule_keys", "radix_sort", "bs_plan", "chain_ranges", "RadixSort", "onesweep", "sched_sort")):
                sched.append((s, e, name))
    gram.sort()
    sched.sort()
    gram = np.array(gram[args.skip:])
    period = np.diff(gram) / 1e3
    busy = np.zeros(len(period))
    for s, e, _ in sched:
        i0 = np.searchsorted(gram, s, side="right") - 1
        i1 = np.searchsorted(gram, e, side="right") - 1
        for i in range(max(i0, 0), min(i1, len(period) - 1) + 1):
            lo, hi = max(s, gram[i]), min(e, gram[i + 1])
            if hi > lo:
                busy[i] += (hi - lo) / 1e3
    quiet = period[busy == 0]
    loud = period[busy > 0]
    print("iterations %d; period mean %.2f us, median %.2f" % (len(period), period.mean(), np.median(period)))
    print("  no schedule kernel inside: %5d iterations, mean %.2f us (median %.2f)" % (len(quiet), quiet.mean(), np.median(quiet)))
    if len(loud):
        print("  schedule kernels inside:   %5d iterations, mean %.2f us (median %.2f), schedule-kernel time inside them %.1f us on average"
              % (len(loud), loud.mean(), np.median(loud), busy[busy > 0].mean()))
        print("  excess over the quiet mean: %.1f us per batch of 64 iterations = %.2f us per iteration"
              % ((loud.mean() - quiet.mean()) * len(loud) / max(len(period) / 64.0, 1e-9), (loud.mean() - quiet.mean()) * len(loud) / len(period)))
    # by kind of schedule kernel: iterations that contain (part of) one of that kind
    kinds = {"keys": ("schedule_keys",), "sort": ("radix_sort", "RadixSort", "onesweep", "sched_sort"), "ranges / plan": ("chain_ranges", "bs_plan")}
    for kind, pats in kinds.items():
        mark = np.zeros(len(period), dtype=bool)
        t_kind = 0.0
        for s, e, name in sched:
            if not any(p in name for p in pats) or s < gram[0]:
                continue
            t_kind += (e - s) / 1e3
            i0 = max(np.searchsorted(gram, s, side="right") - 1, 0)
            i1 = min(np.searchsorted(gram, e, side="right") - 1, len(period) - 1)
            mark[i0:i1 + 1] = True
        if mark.any() and len(quiet):
            print("  %-13s in %5d iterations, mean %.2f us (quiet %.2f): +%.1f us per batch; the kind's GPU time %.1f us per batch"
                  % (kind, mark.sum(), period[mark].mean(), quiet.mean(), (period[mark].mean() - quiet.mean()) * mark.sum() / (len(period) / 64.0),
                     t_kind / (len(period) / 64.0)))
    total = sum(e - s for s, e, _ in sched if s >= gram[0]) / 1e3
    print("  schedule kernels: %.1f us of GPU time per iteration" % (total / len(period)))


if __name__ == "__main__":
    main()
