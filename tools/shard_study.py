#!/usr/bin/env python3
"""How should N user shards reconcile their item factors?  Emulates N ranks on ONE GPU (N engines, one device,
shards run one after the other; the exchange arithmetic is done here with torch ops -- a study tool, not the
product path, which uses one process per GPU and an RCCL all-reduce) and compares the test RMSE after the same
number of iterations with the unsharded run, for merge in {mean, sum, weighted} and several sync periods.

usage: tools/shard_study.py [--workload ml-20m] [--factors 100] [--iters 1000] [--shards 2,8] [--sync 16,115]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="ml-20m")
    ap.add_argument("--factors", type=int, default=100)
    ap.add_argument("--iters", type=int, default=1000)
    ap.add_argument("--shards", default="2,8")
    ap.add_argument("--sync", default="16,115")
    ap.add_argument("--mode", default="hogwild")
    ap.add_argument("--merges", default="mean,sum,weighted")
    ap.add_argument("--gamma", default="", help="adaptive merge, exploratory form: contraction per update of an item row; comma list "
                    "(phi(n) = 1 - (1 - gamma)^n with n = rate x sync).  Empty: the driver's form, phi(r) = 1 - exp(-6 r)")
    ap.add_argument("--sequential", action="store_true",
                    help="also run the unsharded set in ordered mode (= mf_sequential.cu's result, bit for bit) as the reference")
    args = ap.parse_args()
    import torch

    import bench
    import cu2rec_amd as cu
    from cu2rec_amd.engine import DeviceRatings, Engine
    from cu2rec_amd.sharded import plan_users

    train, test = bench.load_dataset(args.workload, 20240917, 0, lambda: None)
    f, hyper = args.factors, (0.01, 0.02, 0.02, 0.02, 0.02)
    dev = torch.device("cuda", 0)
    P0 = cu.initialize_normal_array(train.rows * f, f).reshape(train.rows, f)
    ub0 = cu.initialize_normal_array(train.rows, f)

    def test_rmse(engines, bounds):
        sa = ss = 0.0
        for e, (u0, u1) in zip(engines, bounds):
            out = e.loss(DeviceRatings(test.slice_users(u0, u1), dev))
            sa, ss = sa + out["sum_abs"], ss + out["sum_sq"]
        return (ss / test.nnz) ** 0.5

    results = []
    base = Engine(train.rows, train.cols, f, train.global_bias, device=dev)
    base.sgd(DeviceRatings(train, dev), hyper, 42, 0, args.iters, args.mode)
    ref = test_rmse([base], [(0, train.rows)])
    results.append({"shards": 1, "rmse": ref})
    print("N=1 rmse %.5f" % ref, flush=True)
    del base
    if args.sequential:
        seq = Engine(train.rows, train.cols, f, train.global_bias, device=dev)
        seq.sgd(DeviceRatings(train, dev), hyper, 42, 0, args.iters, "ordered")
        r = test_rmse([seq], [(0, train.rows)])
        results.append({"shards": 1, "mode": "ordered (sequential semantics)", "rmse": r})
        print("N=1 sequential (ordered mode) rmse %.5f (%+.5f vs N=1 %s)" % (r, r - ref, args.mode), flush=True)
        del seq
    for n in [int(v) for v in args.shards.split(",")]:
        b = plan_users(train.rows, n)
        bounds = list(zip(b[:-1], b[1:]))
        shards = [train.slice_users(u0, u1) for u0, u1 in bounds]
        rates = np.stack([cu.api.item_update_rates(s) for s in shards])
        tot = rates.sum(0)
        weights = torch.tensor(np.where(tot > 0, rates / np.maximum(tot, 1e-300), 1.0 / n), dtype=torch.float32, device=dev)
        for sync in [int(v) for v in args.sync.split(",")]:
            merges = []
            for merge in args.merges.split(","):
                merges += [(merge, None)] if merge != "adaptive" else ([("adaptive", float(g)) for g in args.gamma.split(",")] if args.gamma
                                                                       else [("adaptive", 0.0)])
            for merge, gamma in merges:
                if merge == "adaptive":
                    # every shard's delta of an item row is (to first order) the progress phi(n_k) = 1 - (1 - gamma)^n_k of its
                    # n_k updates towards a common target; all N shards' updates in sequence make phi(sum n_k): the SUM of the
                    # deltas is scaled by phi(n_tot) / sum_k phi(n_k) -- 1 for rarely updated items (a plain sum), 1 / N for
                    # items every shard saturates within a period (the mean)
                    n_k = rates * sync if gamma > 0 else rates
                    phi = (lambda x: -np.expm1(x * np.log1p(-gamma))) if gamma > 0 else (lambda x: -np.expm1(-6.0 * x))
                    alpha = np.where(tot > 0, phi(n_k.sum(0)) / np.maximum(phi(n_k).sum(0), 1e-300), 1.0)
                    alpha_t = torch.tensor(alpha, dtype=torch.float32, device=dev)
                engines = [Engine(u1 - u0, train.cols, f, train.global_bias, P=P0[u0:u1], user_bias=ub0[u0:u1], device=dev)
                           for u0, u1 in bounds]
                d = [DeviceRatings(s, dev) for s in shards]
                Qb, ibb = engines[0].Q.clone(), engines[0].item_bias.clone()
                it = 0
                if merge.startswith("twotier"):
                    # exploratory: the items every shard updates all the time (total rate >= 1) are reconciled every `fast`
                    # iterations (their deltas scaled as in the adaptive merge), everything every `sync` iterations
                    fast = int(merge[len("twotier"):] or 8)
                    hot = torch.tensor(tot >= 1.0, device=dev)
                    phi = lambda x: -np.expm1(-6.0 * x)
                    alpha_t = torch.tensor(np.where(tot > 0, phi(tot) / np.maximum(phi(rates).sum(0), 1e-300), 1.0), dtype=torch.float32, device=dev)
                    nh = int(hot.sum().item())
                    while it < args.iters:
                        k = min(fast, args.iters - it, sync - (it % sync))
                        for e, dr, (u0, _) in zip(engines, d, bounds):
                            e.sgd(dr, hyper, 42, it, k, args.mode, True, u0)
                        it += k
                        full = it % sync == 0 or it == args.iters
                        sel = torch.ones_like(hot) if full else hot
                        dQ = torch.stack([e.Q[:train.cols] - Qb[:train.cols] for e in engines]).sum(0)
                        dib = torch.stack([e.item_bias[:train.cols] - ibb[:train.cols] for e in engines]).sum(0)
                        Qb[:train.cols][sel] += (alpha_t[:, None] * dQ)[sel]
                        ibb[:train.cols][sel] += (alpha_t * dib)[sel]
                        for e in engines:
                            e.Q[:train.cols][sel] = Qb[:train.cols][sel]
                            e.item_bias[:train.cols][sel] = ibb[:train.cols][sel]
                    r = test_rmse(engines, bounds)
                    results.append({"shards": n, "sync_every": sync, "merge": merge, "hot_items": nh, "rmse": r, "delta_vs_n1": r - ref})
                    print("N=%d sync=%d merge=%-8s (%d hot items) rmse %.5f (%+.5f vs N=1)" % (n, sync, merge, nh, r, r - ref), flush=True)
                    del engines, d
                    continue
                while it < args.iters:
                    k = min(sync, args.iters - it)
                    for e, dr, (u0, _) in zip(engines, d, bounds):
                        e.sgd(dr, hyper, 42, it, k, args.mode, True, u0)
                    it += k
                    dQ = torch.stack([e.Q - Qb for e in engines])
                    dib = torch.stack([e.item_bias - ibb for e in engines])
                    if merge == "mean":
                        Qb, ibb = Qb + dQ.mean(0), ibb + dib.mean(0)
                    elif merge == "sum":
                        Qb, ibb = Qb + dQ.sum(0), ibb + dib.sum(0)
                    elif merge == "adaptive":
                        Qb = Qb + alpha_t[:, None] * dQ[:, :train.cols].sum(0)
                        ibb = ibb + alpha_t * dib[:, :train.cols].sum(0)
                    else:
                        Qb = Qb + (weights[:, :, None] * dQ[:, :train.cols]).sum(0)
                        ibb = ibb + (weights * dib[:, :train.cols]).sum(0)
                    for e in engines:
                        e.Q.copy_(Qb)
                        e.item_bias.copy_(ibb)
                r = test_rmse(engines, bounds)
                name = merge if not gamma else "%s(gamma=%g)" % (merge, gamma)
                results.append({"shards": n, "sync_every": sync, "merge": name, "rmse": r, "delta_vs_n1": r - ref})
                print("N=%d sync=%d merge=%-8s rmse %.5f (%+.5f vs N=1)" % (n, sync, name, r, r - ref), flush=True)
                del engines, d
    print(json.dumps({"workload": args.workload, "f": f, "iters": args.iters, "mode": args.mode, "results": results}))


if __name__ == "__main__":
    main()
