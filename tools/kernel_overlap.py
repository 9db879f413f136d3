#!/usr/bin/env python3
"""Per kernel class of a rocprofv3 --kernel-trace CSV: mean duration of the launches that overlap a kernel matching PATTERN (e.g. the
schedule's sort) against the launches that do not -- which phase of an iteration pays for running beside it.
usage: kernel_overlap.py <kernel_trace.csv> <pattern>"""
import csv
import sys

import numpy as np

CLASSES = ["bs_gram", "bs_solve", "bs_update", "sgd_ordered", "bs_gate", "bs_signal", "schedule_keys", "sched_sort", "chain_ranges", "bs_plan"]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    pat = sys.argv[2]
    ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows]
    marks = sorted((s, e) for s, e, n in ev if pat in n)
    starts = np.array([m[0] for m in marks])
    ends = np.array([m[1] for m in marks])
    print("%d launches of '%s', mean %.1f us" % (len(marks), pat, np.mean(ends - starts) / 1e3 if len(marks) else 0))
    for c in CLASSES:
        a, b = [], []
        for s, e, n in ev:
            if c not in n:
                continue
            i = np.searchsorted(starts, e) - 1  # last mark starting before this kernel ends
            over = i >= 0 and ends[i] > s
            (a if over else b).append((e - s) / 1e3)
        if a or b:
            print("  %-14s beside: n %5d mean %8.1f us   alone: n %5d mean %8.1f us" % (c, len(a), np.mean(a) if a else 0, len(b), np.mean(b) if b else 0))


if __name__ == "__main__":
    main()
