#!/usr/bin/env python3
"""Median timeline of one block-solve iteration from a rocprofv3 --kernel-trace CSV (…_kernel_trace.csv): start and end of
each kernel of the iteration relative to the start of its phase 1, over the iterations whose period is below `cap` us
(the ones not beside a schedule kernel of the next batch), and the mean / median period over all of them.

    python tools/iteration_timeline.py <kernel_trace.csv> [cap_us=100]
"""
import csv
import statistics as st
import sys

KEYS = ["bs_gram", "bs_solve", "bs_update_pipe", "bs_update", "bs_gate", "bs_signal", "sgd_ordered", "sgd_walk"]


def short(name):
    for k in KEYS:
        if k in name:
            return k
    return None


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    cap = float(sys.argv[2]) if len(sys.argv) > 2 else 100.0
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
    its, cur = [], None
    for s, e, n in ev:
        k = short(n)
        if k == "bs_gram":
            if cur:
                its.append(cur)
            cur = {"t0": s, "bs_gram": [(s, e)]}
        elif cur and k:
            cur.setdefault(k, []).append((s, e))
    per = [(b["t0"] - a["t0"]) / 1e3 for a, b in zip(its, its[1:])]
    good = [i for i, p in enumerate(per) if p < cap]
    print("iterations %d  period mean %.1f median %.1f us;  %d below %.0f us: mean %.1f median %.1f" %
          (len(per), st.mean(per), st.median(per), len(good), cap, st.mean(per[i] for i in good), st.median(per[i] for i in good)))
    for k in KEYS:
        for j in range(3):
            sel = [its[i][k][j] for i in good if k in its[i] and len(its[i][k]) > j]
            if len(sel) < len(good) // 2:
                continue
            t0s = [its[i]["t0"] for i in good if k in its[i] and len(its[i][k]) > j]
            s = st.median((a[0] - t) / 1e3 for a, t in zip(sel, t0s))
            e = st.median((a[1] - t) / 1e3 for a, t in zip(sel, t0s))
            print("  %-16s #%d  start %6.1f  end %6.1f  (%.1f)" % (k, j, s, e, e - s))


if __name__ == "__main__":
    main()
