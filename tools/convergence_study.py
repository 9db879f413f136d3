#!/usr/bin/env python3
"""Test-RMSE trajectory of the GPU SGD modes on the same data and the same sampler stream.  `ordered` IS the
sequential (mf_sequential.cu) trajectory -- the parity tests show it bit-identical to the CPU oracle -- so it serves
as the reference curve here; this tool itself never touches oracle/.

usage: tools/convergence_study.py [--workload ml-1m] [--factors 50] [--iters 2000] [--every 250]
                                  [--modes hogwild,hogwild-streaming,ordered]
       hogwild = the default policy (resident launches when the rows fit), hogwild-streaming = one launch per iteration
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="ml-1m")
    ap.add_argument("--factors", type=int, default=50)
    ap.add_argument("--iters", type=int, default=2000)
    ap.add_argument("--every", type=int, default=250)
    ap.add_argument("--modes", default="hogwild,ordered")
    ap.add_argument("--lr", type=float, default=0.01)
    args = ap.parse_args()
    import bench
    import cu2rec_amd as cu
    train, test = bench.load_dataset(args.workload, 20240917, 0, lambda: None)
    f, hyper = args.factors, (args.lr, 0.02, 0.02, 0.02, 0.02)
    d_tr, d_te = cu.DeviceCSR(train), cu.DeviceCSR(test)
    out = {"workload": args.workload, "f": f, "hyper": hyper, "iters": args.iters, "curves": {}}
    for mode in args.modes.split(","):
        label = mode
        prev_policy = cu.lib().cu2rec_hogwild_resident(-1)
        if mode == "hogwild-streaming":  # one launch per iteration, user rows through HBM (CU2REC_RESIDENT=0)
            mode = "hogwild"
            cu.lib().cu2rec_hogwild_resident(0)
        model = cu.Model(train.rows, train.cols, f, train.global_bias)
        curve, t_sgd = [(0, model.loss(d_te)["rmse"], model.loss(d_tr)["rmse"])], 0.0
        for it in range(0, args.iters, args.every):
            t0 = time.perf_counter()
            model.sgd(d_tr, hyper, 42, it, args.every, mode=mode)
            r = model.loss(d_te)
            t_sgd += time.perf_counter() - t0
            curve.append((it + args.every, r["rmse"], model.loss(d_tr)["rmse"]))
        cu.lib().cu2rec_hogwild_resident(prev_policy)
        mode = label
        out["curves"][mode] = {"points": curve, "seconds": t_sgd}
        print(mode, "%.2fs" % t_sgd, " ".join("%d:%.5f" % (a, b) for a, b, _ in curve), flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
