#!/usr/bin/env python3
"""Timeline of one block-solve iteration from a rocprofv3 kernel trace (`rocprofv3 --kernel-trace --output-format csv`).
For the n-th launch of each kernel, start and end relative to the start of the n-th bs_gram_kernel; mean / p10 / p90 over the
iterations [--skip, --skip + --count).  usage: kernel_timeline.py <kernel_trace.csv> [--skip N] [--count M]"""
import argparse
import csv
import collections

import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--skip", type=int, default=100)
    ap.add_argument("--count", type=int, default=200)
    args = ap.parse_args()
    by_name = collections.defaultdict(list)
    with open(args.trace) as fh:
        for row in csv.DictReader(fh):
            name = row["Kernel_Name"]
            short = next((k for k in ("bs_gram", "bs_solve", "bs_update", "sgd_ordered", "sgd_walk", "bs_plan", "schedule_keys", "RadixSort",
                                      "Onesweep", "Histogram") if k in name), None)
            if short:
                by_name[short].append((int(row["Start_Timestamp"]), int(row["End_Timestamp"])))
    for k in by_name:
        by_name[k].sort()
    gram = by_name.get("bs_gram", [])
    n = min(len(v) for k, v in by_name.items() if k in ("bs_gram", "bs_solve", "bs_update", "sgd_ordered"))
    lo, hi = min(args.skip, max(n - 2, 0)), min(args.skip + args.count, n - 1)
    print("iterations %d..%d of %d; times in us relative to the start of the iteration's bs_gram_kernel" % (lo, hi, n))
    period = np.diff([g[0] for g in gram[lo:hi + 1]]) / 1e3
    print("period (gram start to next gram start): mean %.2f  p10 %.2f  p50 %.2f  p90 %.2f  max %.2f" %
          (period.mean(), np.percentile(period, 10), np.percentile(period, 50), np.percentile(period, 90), period.max()))
    for k in ("bs_solve", "bs_gram", "sgd_ordered", "sgd_walk", "bs_update"):
        if k not in by_name or len(by_name[k]) < hi:
            continue
        st = np.array([by_name[k][i][0] - gram[i][0] for i in range(lo, hi)]) / 1e3
        en = np.array([by_name[k][i][1] - gram[i][0] for i in range(lo, hi)]) / 1e3
        print("%-12s start mean %7.2f (p10 %7.2f p90 %7.2f)   end mean %7.2f (p10 %7.2f p90 %7.2f)   duration mean %6.2f" %
              (k, st.mean(), np.percentile(st, 10), np.percentile(st, 90), en.mean(), np.percentile(en, 10), np.percentile(en, 90),
               (en - st).mean()))


if __name__ == "__main__":
    main()
