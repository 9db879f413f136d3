#!/usr/bin/env python3
"""Average duration per kernel from a rocprofv3 --kernel-trace --stats --output-format csv run: kernel_avgs.py DIR"""
import csv
import glob
import sys

for f in glob.glob(sys.argv[1] + "/*/*kernel_stats.csv"):
    for r in list(csv.DictReader(open(f)))[:int(sys.argv[2]) if len(sys.argv) > 2 else 8]:
        print("%-60s calls %6s avg %10.1f us  %5s%%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"][:5]))
