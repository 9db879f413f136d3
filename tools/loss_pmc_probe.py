#!/usr/bin/env python3
"""The fused loss pass over the TRAIN set of a BASELINE shape, alone, for a rocprofv3 --pmc pass (tools/pmc_passes.sh with
PMC_PROGRAM=tools/loss_pmc_probe.py): a model 200 Hogwild iterations into training (so that the factors are not the initialisation's
near-zeros), then `--calls` calls of cu2rec_loss on the train set ONLY (the summary averages per kernel name: a test-set call would mix in).
usage: tools/loss_pmc_probe.py [--workload ml-20m --factors 100 --calls 5]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import cu2rec_amd as cu

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="ml-20m")
ap.add_argument("--factors", type=int, default=100)
ap.add_argument("--calls", type=int, default=5)
args = ap.parse_args()
tr, te = bench.load_dataset(args.workload, 20240917, 0, lambda: None)
d_tr, d_te = cu.DeviceCSR(tr), cu.DeviceCSR(te)
m = cu.Model(tr.rows, tr.cols, args.factors, tr.global_bias)
m.sgd(d_tr, (0.01, 0.02, 0.02, 0.02, 0.02), 42, 0, 200, mode="hogwild")
for _ in range(args.calls):
    r = m.loss(d_tr)
print("train", tr.nnz, r["rmse"])
