#!/usr/bin/env python3
"""Microseconds per SGD iteration for the first 1/N of a named set's users (Hogwild: default policy, resident launches
where they pay): what ONE GPU of an N-GPU strong-scaling run does between two exchanges.
  python tools/shard_size_probe.py [--workload ml-20m --factors 100 --shards 1,2,4,8 --iters 500 --mode blocksolve --rates 60,120,240]"""
import argparse
import ctypes
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
    ap.add_argument("--shards", default="1,2,4,8")
    ap.add_argument("--iters", type=int, default=500)
    ap.add_argument("--mode", default="hogwild", help="hogwild | blocksolve | ordered")
    ap.add_argument("--rates", default="", help="block-solve thresholds to try, comma separated (default: the library's)")
    args = ap.parse_args()
    import torch
    import bench
    import cu2rec_amd as cu
    from cu2rec_amd.engine import DeviceRatings, Engine
    from cu2rec_amd.sharded import plan_users
    train, _ = bench.load_dataset(args.workload, 20240917, 0, lambda: None)
    dev = torch.device("cuda", 0)
    hyper = (0.01, 0.02, 0.02, 0.02, 0.02)
    for n in [int(v) for v in args.shards.split(",")]:
        b = plan_users(train.rows, n)
        shard = train.slice_users(b[0], b[1])
        users = int(np.count_nonzero(np.diff(shard.indptr)))
        d = DeviceRatings(shard, dev)
        eng = Engine(shard.rows, shard.cols, args.factors, shard.global_bias, device=dev)
        blocks, rows = ctypes.c_int(0), ctypes.c_int(0)
        resident = cu.lib().cu2rec_hogwild_resident_plan(shard.rows, args.factors, args.iters, ctypes.byref(blocks), ctypes.byref(rows))
        if args.mode != "hogwild":
            for rate in [float(v) for v in args.rates.split(",") if v] or [0.0]:
                if rate > 0:
                    cu.api.blocksolve_min_rate(rate)
                d = DeviceRatings(shard, dev)  # the schedule (and its hot set) is made with the ratings object
                eng.sgd(d, hyper, 42, 0, 64, mode=args.mode)
                torch.cuda.synchronize()
                best = 1e9
                for rep in range(3):
                    t0 = time.perf_counter()
                    eng.sgd(d, hyper, 42, 64 + rep * args.iters, args.iters, mode=args.mode)
                    torch.cuda.synchronize()
                    best = min(best, time.perf_counter() - t0)
                print("1/%d of %s: %d users, %s (hot threshold %s): %.2f us per iteration = %.3e updates/s per GPU, x%d = %.3e" % (
                    n, args.workload, users, args.mode, rate or "default", 1e6 * best / args.iters, users * args.iters / best, n,
                    n * users * args.iters / best), flush=True)
            continue
        eng.sgd(d, hyper, 42, 0, args.iters)
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(4):
            t0 = time.perf_counter()
            eng.sgd(d, hyper, 42, (rep + 1) * args.iters, args.iters)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        print("1/%d of %s: %d users, %s: %.2f us per iteration = %.3e updates/s per GPU, x%d = %.3e" % (
            n, args.workload, users, ("resident, %d workgroups x %d rows per group" % (blocks.value, rows.value)) if resident == 1
            else "streaming", 1e6 * best / args.iters, users * args.iters / best, n, n * users * args.iters / best), flush=True)


if __name__ == "__main__":
    main()
