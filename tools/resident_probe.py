#!/usr/bin/env python3
"""Microseconds per Hogwild iteration of the resident launch vs the streaming launches, same data (GPU box).
  python tools/resident_probe.py [--workload ml-20m --factors 100 --iters 115 --reps 6]
With CU2REC_AMD_LIB pointing at a tools/build_variant.sh build this times ablations."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="ml-20m")
    ap.add_argument("--factors", type=int, default=100)
    ap.add_argument("--iters", type=int, default=115)
    ap.add_argument("--reps", type=int, default=6)
    ap.add_argument("--policies", default="0,2")
    args = ap.parse_args()
    import torch
    import bench
    import cu2rec_amd as cu
    from cu2rec_amd.engine import DeviceRatings, Engine
    train, test = bench.load_dataset(args.workload, 20240917, 0, lambda: None)
    dev = torch.device("cuda", 0)
    hyper = (0.01, 0.02, 0.02, 0.02, 0.02)
    users = int(np.count_nonzero(np.diff(train.indptr)))
    d_train, d_test = DeviceRatings(train, dev), DeviceRatings(test, dev)
    for policy in [int(p) for p in args.policies.split(",")]:
        cu.lib().cu2rec_hogwild_resident(policy)
        eng = Engine(train.rows, train.cols, args.factors, train.global_bias, device=dev)
        it = 0
        eng.sgd(d_train, hyper, 42, it, args.iters)
        it += args.iters
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(args.reps):
            t0 = time.perf_counter()
            eng.sgd(d_train, hyper, 42, it, args.iters)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
            it += args.iters
        print("lib %s policy %d: %.2f us/iteration (%.3e updates/s), test rmse after %d iterations %.5f" % (
            os.environ.get("CU2REC_AMD_LIB", "default"), policy, 1e6 * best / args.iters, users * args.iters / best, it,
            eng.loss(d_test)["rmse"]), flush=True)


if __name__ == "__main__":
    main()
