#!/usr/bin/env python3
"""Development probe for CU2REC_SGD_BLOCKSOLVE on the GPU box: small-set parity against the oracle, then speed and the
gap to the ordered mode (= the sequential result, bit for bit) on a named shape.
  python tools/blocksolve_probe.py [--workload ml-20m] [--factors 100] [--iters 200] [--rate 8]"""
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
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--rate", type=float, default=0.0)
    ap.add_argument("--skip-small", action="store_true")
    ap.add_argument("--skip-big", action="store_true")
    ap.add_argument("--stamps", action="store_true", help="wavefront timelines of one block-solve iteration")
    args = ap.parse_args()
    import cu2rec_amd as cu
    from cu2rec_amd import api, synth
    from oracle import oracle as orc
    import bench
    hyper = (0.01, 0.02, 0.02, 0.02, 0.02)
    if not args.skip_small:
        for users, items, nnz, f, iters, rate in ((300, 120, 6000, 10, 5, 2.0), (300, 120, 6000, 100, 70, 2.0),
                                                  (3000, 40, 30000, 100, 6, 1.0), (3000, 40, 30000, 50, 6, 1e9),
                                                  (2000, 300, 40000, 128, 5, 0.5), (2000, 300, 40000, 200, 5, 0.5),
                                                  (2000, 300, 40000, 252, 3, 4.0), (500, 50, 5000, 8, 10, 0.01)):
            api.blocksolve_min_rate(rate)
            tr, te = synth.make_ratings(users, items, nnz, min_degree=3, seed=users + f)
            P, Q, ub, ib = orc.init_model(tr.rows, tr.cols, f)
            model = cu.Model(tr.rows, tr.cols, f, tr.global_bias)
            model.sgd(cu.DeviceCSR(tr), hyper, 42, 0, iters, mode="blocksolve")
            orc.sgd_iterations(orc.CSR(tr.indptr, tr.indices, tr.data, tr.rows, tr.cols, tr.global_bias), P, Q, ub, ib,
                               tr.global_bias, hyper, 42, 0, iters, dot_order=orc.DOT_TREE16)
            diffs = [float(np.abs(g.astype(np.float64) - w).max()) for g, w in zip(model.download(), (P, Q, ub, ib))]
            print("small %5d x %4d f=%3d iters=%3d rate=%g  max|diff| P %.2e Q %.2e ub %.2e ib %.2e" %
                  ((users, items, f, iters, rate) + tuple(diffs)), flush=True)
    if args.skip_big:
        return
    if args.rate > 0:
        api.blocksolve_min_rate(args.rate)
    tr, te = bench.load_dataset(args.workload, 20240917, 0, lambda: None)
    users = int(np.count_nonzero(np.diff(tr.indptr)))
    f = args.factors
    d_tr, d_te = cu.DeviceCSR(tr), cu.DeviceCSR(te)
    if args.stamps:
        import torch
        from cu2rec_amd._lib import check, lib
        cap = 1 << 18
        buf = torch.zeros(1 + 8 * cap, dtype=torch.int64, device="cuda")
        model = cu.Model(tr.rows, tr.cols, f, tr.global_bias)
        model.sgd(d_tr, hyper, 42, 0, 66, mode="blocksolve")
        torch.cuda.synchronize()
        check(lib().cu2rec_debug_blocksolve_stamps(buf.data_ptr(), cap))
        model.sgd(d_tr, hyper, 42, 66, 1, mode="blocksolve")
        torch.cuda.synchronize()
        check(lib().cu2rec_debug_blocksolve_stamps(None, 0))
        h = buf.cpu().numpy()
        rec = h[1:1 + 8 * cap].reshape(cap, 8)
        rec = rec[rec[:, 0] > 0]
        xcc = rec[:, 0] >> 32
        rec[:, 0] &= 0xffffffff
        for kid, nm in ((2, "solver"), (6, "build")):
            sel = rec[:, 0] == kid
            print("xcc of", nm, "records (id -> xcc):", [(int(i), int(x)) for i, x in zip(rec[sel, 1][:24], xcc[sel][:24])])
        n = len(rec)
        t_min = rec[:, 2].min()
        names = {1: "gram", 2: "solver", 3: "loader", 4: "update", 5: "walk", 6: "build", 7: "cross"}
        print("stamps: %d wavefront records; times in us from the first start" % n)
        for kid in sorted(names):
            r = rec[rec[:, 0] == kid]
            if not len(r):
                continue
            st, en = (r[:, 2] - t_min) / 100.0, (r[:, 3] - t_min) / 100.0
            dur = en - st
            print("%-7s waves %6d  start %7.2f..%7.2f  end %7.2f..%7.2f  dur mean %6.2f p50 %6.2f p90 %6.2f max %6.2f" %
                  (names[kid], len(r), st.min(), st.max(), en.min(), en.max(), dur.mean(), np.percentile(dur, 50),
                   np.percentile(dur, 90), dur.max()))
            if r[:, 4:].any():
                rr = r[(r[:, 4:] > 0).all(axis=1)]
                mk = (rr[:, 4:] - rr[:, 2:3]) / 100.0
                print("   marks after start (mean us over %d waves):" % len(rr), [round(float(v), 2) for v in mk.mean(axis=0)])
                if kid in (2, 3):
                    for row in list(rr[:8]) + list(rr[-4:]):
                        print("     chain %d wave %d:" % (row[1] // 4, row[1] % 4), [round(float(v - row[2]) / 100.0, 2) for v in row[4:]], "end", round(float(row[3] - row[2]) / 100.0, 2))
            if kid in (3, 4) and not os.environ.get("CU2REC_PROBE_FINE"):
                cyc = (r[:, 7] - r[:, 2]).astype(np.float64)
                tick = (r[:, 3] - r[:, 2]).astype(np.float64)
                ok = tick > 0
                ghz = cyc[ok] / (tick[ok] * 10.0)
                print("   shader clock over the wavefront's life: mean %.3f GHz (p10 %.3f, p90 %.3f)" % (ghz.mean(), np.percentile(ghz, 10), np.percentile(ghz, 90)))
            if kid in (1, 4) and os.environ.get("CU2REC_PROBE_FINE") and r[:, 4:].any():
                rr = r[(r[:, 4:] > 0).all(axis=1)]
                mk = (rr[:, 4:] - rr[:, 2:3]) / 100.0
                d = (rr[:, 3] - rr[:, 2]) / 100.0
                for q in (50, 90, 100):
                    print("   fine marks p%d:" % q, [round(float(v), 2) for v in np.percentile(mk, q, axis=0)], "end", round(float(np.percentile(d, q)), 2))
            if kid == 1:
                order = np.argsort(-en)[:6]
                for i in order:
                    print("     last gram waves: id %d start %.2f end %.2f marks" % (r[i, 1], st[i], en[i]), [round(float(v - r[i, 2]) / 100.0, 2) for v in r[i, 4:]])
            if kid == 2:
                order = np.argsort(-dur)[:8]
                print("   longest solver waves (chain, start, end):", [(int(r[i, 1]) // 4, round(float(st[i]), 2), round(float(en[i]), 2)) for i in order])
        return
    res = {}
    for mode in ("blocksolve", "ordered"):
        model = cu.Model(tr.rows, tr.cols, f, tr.global_bias)
        model.sgd(d_tr, hyper, 42, 0, 64, mode=mode)
        model.loss(d_te)
        t0 = time.perf_counter()
        model.sgd(d_tr, hyper, 42, 64, args.iters, mode=mode)
        rm = model.loss(d_te)["rmse"]
        dt = time.perf_counter() - t0
        res[mode] = (model.download(), rm)
        print("%s: %.1f us/iteration = %.3e updates/s, test rmse after %d iterations %.6f" %
              (mode, 1e6 * dt / args.iters, users * args.iters / dt, 64 + args.iters, rm), flush=True)
    (a, ra), (b, rb) = res["blocksolve"], res["ordered"]
    print("blocksolve vs ordered: rmse gap %.2e; max|diff| P %.2e Q %.2e ub %.2e ib %.2e" %
          ((abs(ra - rb),) + tuple(float(np.abs(x.astype(np.float64) - y).max()) for x, y in zip(a, b))), flush=True)


if __name__ == "__main__":
    main()
