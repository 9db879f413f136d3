#!/usr/bin/env python3
"""The north-star sentence on BASELINE.json configs[2], CPU side: the ORACLE's train() (oracle/cu2rec_oracle.c: orc_train =
mf_sequential.cu:102-201 under training.cu:118,146-155's schedule) on the ML-20M shape, f=100, in the reference's OWN arithmetic --
sequential dot (mf_sequential.cu:122-125: DOT_SEQ) and either float loss accumulators (mf_sequential.cu:147-174: ACC_F32, what
the reference prints and decides on) or double ones (loss.cu:185-190: ACC_F64).  Test infrastructure: the product never imports
this; tests/conftest.py runs `run()` on spare host cores behind the GPU suite and tests/test_gpu_blocksolve.py compares
cu2rec_train(CU2REC_SGD_BLOCKSOLVE) with it.

usage: tools/oracle_converged.py [--workload ml-20m --factors 100 --iters 5000] [--dot SEQ|TREE16] [--acc F32|F64] [--out file.json]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(workload, f, iters, dot_name="SEQ", acc_name="F32", seed=42, data=None):
    """-> (record, (P, Q, user_bias, item_bias)).  record: the logged checks (iteration, train/test MAE/RMSE, rate AFTER the
    check's decision), the iterations at which the rate decayed, final rate, wall seconds."""
    import bench
    from oracle import oracle as orc
    tr, te = data if data is not None else bench.load_dataset(workload, 20240917, 0, lambda: None)
    o_tr = orc.CSR(tr.indptr, tr.indices, tr.data, tr.rows, tr.cols, tr.global_bias)
    o_te = orc.CSR(te.indptr, te.indices, te.data, te.rows, te.cols, te.global_bias)
    P, Q, ub, ib = orc.init_model(tr.rows, tr.cols, f)
    cfg = orc.default_config(total_iterations=iters, n_factors=f, learning_rate=0.01, seed=seed, P_reg=0.02, Q_reg=0.02,
                             user_bias_reg=0.02, item_bias_reg=0.02)  # preprocessing/create_config.py:25-32
    t0 = time.perf_counter()
    log = orc.train(o_tr, o_te, cfg, P, Q, ub, ib, tr.global_bias, dot_order=getattr(orc, "DOT_" + dot_name),
                    acc=getattr(orc, "ACC_" + acc_name), schedule=orc.SCHED_PATIENCE)
    wall = time.perf_counter() - t0
    decays, lr = [], np.float32(0.01)
    for e in log:
        if np.float32(e["lr"]) != lr:
            decays.append(e["iteration"])
            lr = np.float32(e["lr"])
    rec = {"workload": workload, "f": f, "iterations": iters, "seed": seed, "dot_order": dot_name, "loss_accumulators": acc_name,
           "schedule": "training.cu:118,146-155: check at 1, every 500, last; patience 2; decay 0.2; lr .01, reg .02",
           "users": tr.rows, "items": tr.cols, "train_nnz": tr.nnz, "test_nnz": te.nnz,
           "checks": log, "decay_iterations": decays, "final_learning_rate": float(cfg.learning_rate),
           "final_test_rmse": log[-1]["test_rmse"], "final_train_rmse": log[-1]["train_rmse"],
           "min_test_rmse": min(e["test_rmse"] for e in log), "seconds_wall": wall, "cores": 1}
    return rec, (P, Q, ub, ib)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="ml-20m")
    ap.add_argument("--factors", type=int, default=100)
    ap.add_argument("--iters", type=int, default=5000)
    ap.add_argument("--dot", default="SEQ")
    ap.add_argument("--acc", default="F32")
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    rec, _ = run(args.workload, args.factors, args.iters, args.dot, args.acc, args.seed)
    for e in rec["checks"]:
        print("%5d  train %.6f  test %.6f  lr %.3g" % (e["iteration"], e["train_rmse"], e["test_rmse"], e["lr"]), flush=True)
    print("decays at %s, final test rmse %.6f, %.1f s" % (rec["decay_iterations"], rec["final_test_rmse"], rec["seconds_wall"]))
    if args.out:
        with open(args.out, "w") as fh:
            json.dump(rec, fh, indent=1)


if __name__ == "__main__":
    main()
