#!/usr/bin/env python3
"""SURVEY.md section 8c, Tier 2 as written: *converged* Hogwild against the sequential result.

Every mode runs the product's own train() (cu2rec_train, csrc/train.cpp) under the reference's schedule -- loss check at
iteration 1, every check_error = 500 iterations and at the end, patience 2, learning-rate decay 0.2 (training.cu:118,146-155,
config.h:41-51) -- for --iters iterations, long enough for the learning rate to have decayed at least three times (the run
reports how often it did; a run with fewer decays is marked not converged).  `ordered` IS mf_sequential.cu's trajectory (the
parity tests show it bit-identical to the CPU oracle), so it is the reference run here; this tool never touches oracle/.

Per (shape, sampler seed, mode): the iterations at which the rate decayed, the minimum and the final test RMSE, the final
rate, and max |dP|, max |dQ|, max |d user_bias|, max |d item_bias| against the sequential run of the same seed.

usage: tools/tier2_converged.py [--workload ml-20m --factors 100] [--iters 8000] [--seeds 42,7,20240917]
                                [--modes ordered,blocksolve,hogwild-resident,hogwild-streaming] [--out file.json]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

NORTH_STAR_TOLERANCE = 1e-4


def replay_schedule(losses, patience, decay, lr0, check_error):
    """The decisions train_schedule_core.hpp took, recomputed from the logged validation RMSEs (training.cu:129,146-155):
    -> (checks [(iteration, rmse)], decay iterations, final rate)."""
    checks = [(i + 1, float(v)) for i, v in enumerate(losses) if np.isfinite(v)]
    last, cur, lr, decays = np.float32(np.finfo(np.float32).max), patience, np.float32(lr0), []
    for it, r in checks:
        if last < np.float32(r):
            cur -= 1
        if cur <= 0:
            cur = patience
            lr = np.float32(lr * np.float32(decay))
            decays.append(it)
        last = np.float32(r)
    return checks, decays, float(lr)


def run_mode(cu, d_tr, d_te, f, seed, iters, label):
    mode, policy = label, None
    if label.startswith("hogwild"):
        mode, policy = "hogwild", {"hogwild-resident": 2, "hogwild-streaming": 0}[label]
    cfg = cu.api.default_config(total_iterations=iters, n_factors=f, learning_rate=0.01, seed=seed, P_reg=0.02, Q_reg=0.02,
                                user_bias_reg=0.02, item_bias_reg=0.02)  # preprocessing/create_config.py:25-32
    prev = cu.lib().cu2rec_hogwild_resident(policy) if policy is not None else None
    t0 = time.perf_counter()
    try:
        P, Q, losses, ub, ib, stats = cu.api.train(d_tr, d_te, cfg, mode=mode, verbose=False, return_stats=True)
    finally:
        if prev is not None:
            cu.lib().cu2rec_hogwild_resident(prev)
    wall = time.perf_counter() - t0
    checks, decays, lr = replay_schedule(losses, int(cfg.patience), float(cfg.learning_rate_decay), 0.01, int(cfg.check_error))
    assert abs(lr - float(cfg.learning_rate)) <= 1e-12 + 1e-6 * lr, (lr, float(cfg.learning_rate))  # the replay IS the run's schedule
    rec = {"mode": label, "seed": seed, "iterations": iters, "decay_iterations": decays, "n_decays": len(decays),
           "final_learning_rate": float(cfg.learning_rate), "converged": len(decays) >= 3 or float(cfg.learning_rate) < 1e-5,
           "min_test_rmse": min(r for _, r in checks), "min_test_rmse_iteration": min(checks, key=lambda c: c[1])[0],
           "final_test_rmse": checks[-1][1], "final_train_rmse": float(stats.last_train_rmse),
           "checks": checks, "seconds_sgd": float(stats.seconds_sgd), "seconds_wall": wall}
    return rec, (P, Q, ub, ib)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="ml-20m")
    ap.add_argument("--factors", type=int, default=100)
    ap.add_argument("--iters", type=int, default=8000)
    ap.add_argument("--seeds", default="42,7,20240917")
    ap.add_argument("--modes", default="ordered,blocksolve,hogwild-resident,hogwild-streaming")
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    import bench
    import cu2rec_amd as cu
    train, test = bench.load_dataset(args.workload, 20240917, 0, lambda: None)
    f = args.factors
    d_tr, d_te = cu.DeviceCSR(train), cu.DeviceCSR(test)
    modes = args.modes.split(",")
    assert modes[0] == "ordered", "the sequential run comes first: the others are compared with it"
    out = {"workload": args.workload, "f": f, "users": train.rows, "items": train.cols, "train_nnz": train.nnz, "test_nnz": test.nnz,
           "schedule": "training.cu:118,146-155: check at 1, every 500, last; patience 2; decay 0.2; lr .01, reg .02",
           "tolerance": NORTH_STAR_TOLERANCE, "runs": [], "summary": {}}
    for seed in (int(s) for s in args.seeds.split(",")):
        seq_rec, seq_par = None, None
        for label in modes:
            if label == "hogwild-resident" and cu.lib().cu2rec_hogwild_resident_plan(train.rows, f, 500, None, None) != 1:
                continue
            rec, par = run_mode(cu, d_tr, d_te, f, seed, args.iters, label)
            if label == "ordered":
                seq_rec, seq_par = rec, par
            rec["gap_final_test_rmse"] = abs(rec["final_test_rmse"] - seq_rec["final_test_rmse"])
            rec["gap_min_test_rmse"] = abs(rec["min_test_rmse"] - seq_rec["min_test_rmse"])
            rec["same_decay_iterations_as_sequential"] = rec["decay_iterations"] == seq_rec["decay_iterations"]
            for name, a, b in zip(("max_abs_dP", "max_abs_dQ", "max_abs_d_user_bias", "max_abs_d_item_bias"), par, seq_par):
                rec[name] = float(np.abs(a.astype(np.float64) - b).max())
            rec["within_tolerance"] = bool(rec["gap_final_test_rmse"] <= NORTH_STAR_TOLERANCE)
            out["runs"].append(rec)
            print("%-8s f=%d seed %-9d %-18s decays at %s  lr %.3g  min %.6f @%d  final %.6f  gap %.2e  |dP| %.2e |dQ| %.2e  %.1fs"
                  % (args.workload, f, seed, label, rec["decay_iterations"], rec["final_learning_rate"], rec["min_test_rmse"],
                     rec["min_test_rmse_iteration"], rec["final_test_rmse"], rec["gap_final_test_rmse"], rec["max_abs_dP"],
                     rec["max_abs_dQ"], rec["seconds_wall"]), flush=True)
    for label in modes:
        runs = [r for r in out["runs"] if r["mode"] == label]
        if runs:
            out["summary"][label] = {"seeds": len(runs), "all_converged": all(r["converged"] for r in runs),
                                     "max_gap_final_test_rmse": max(r["gap_final_test_rmse"] for r in runs),
                                     "gaps_final_test_rmse": [r["gap_final_test_rmse"] for r in runs],
                                     "max_gap_min_test_rmse": max(r["gap_min_test_rmse"] for r in runs),
                                     "within_tolerance_on_every_seed": all(r["within_tolerance"] for r in runs),
                                     "final_test_rmse": [r["final_test_rmse"] for r in runs]}
    print(json.dumps(out["summary"], indent=1))
    if args.out:
        with open(args.out, "w") as fh:
            json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()
