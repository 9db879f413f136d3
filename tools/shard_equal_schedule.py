#!/usr/bin/env python3
"""What a user-sharded run's converged gap to N = 1 is MADE of: merge error, or the reference's patience rule
taking its decisions at other checks?  Per shape and N (the PRODUCT's driver, N ranks as threads of one process on one GPU: the
in-process world of tests/test_gpu_sharded.py):
  own      cu2rec_train_sharded: the sharded run takes its own patience decisions on its own global test RMSE (training.cu:146-155);
  forced   cu2rec_shard_job_run segment by segment with the N = 1 run's learning-rate history (its decay iterations replayed as a
           given schedule), exchange + global test loss behind every segment: EQUAL LR histories, so what is left is the merge.
Printed per run: per-check gap to the N = 1 run, the largest of them, end-point gap, where the own schedule decayed.
usage: tools/shard_equal_schedule.py [workload factors iterations shards,...] [--merge adaptive] [--sync 0]"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import test_gpu_sharded as t

args = [a for a in sys.argv[1:] if not a.startswith("--")]
opts = {a.split("=")[0].lstrip("-"): a.split("=")[1] for a in sys.argv[1:] if a.startswith("--") and "=" in a}
wl, f, its = (args[0], int(args[1]), int(args[2])) if len(args) >= 3 else ("ml-20m", 100, 8000)
shards = tuple(int(v) for v in args[3].split(",")) if len(args) >= 4 else (2, 4, 8)
merge, sync = opts.get("merge", "adaptive"), int(opts.get("sync", "0"))

ref, ref_lr = t._unsharded(wl, f, converged_iters=its)
e0 = dict(t._unsharded.last_extra)
decays0 = t.replay_decays(e0["checks"])
forced = t.forced_segments(e0["checks"], decays0)
print("%s f=%d %d iterations, merge %s, sync %s" % (wl, f, its, merge, sync or "epoch"))
print("N=1   final %.6f  min %.6f @%d  decays at %s  lr %.3g" % (ref, e0["min"], e0["at"], decays0, ref_lr), flush=True)
out = {"workload": wl, "f": f, "iterations": its, "merge": merge, "sync_every": sync or "epoch",
       "n1": {"final": ref, "min": e0["min"], "min_at": e0["at"], "decays": decays0, "checks": e0["checks"]}, "runs": {}}
for n in shards:
    rec = {}
    for kind in ("own", "forced"):
        if kind == "own":
            r, ex, same, lr = t._sharded_run(wl, f, n, merge=merge, sync=sync, converged_iters=its)
        else:
            r, ex, same, lr = t._sharded_run(wl, f, n, merge=merge, sync=sync, forced=forced)
        e = t._sharded_run.last_extra
        gaps = [(it, v - v0) for (it, v), (_, v0) in zip(e["checks"], e0["checks"])]
        worst = max(gaps, key=lambda g: abs(g[1]))
        rec[kind] = {"final": r, "gap_final": r - ref, "min": e["min"], "min_at": e["at"], "gap_min": e["min"] - e0["min"], "max_abs_check_gap": abs(worst[1]),
                     "max_abs_check_gap_at": worst[0], "decays": t.replay_decays(e["checks"]) if kind == "own" else decays0, "check_gaps": gaps,
                     "exchanges": ex, "replicas_identical": bool(same)}
        print("N=%d %-6s final %.6f (gap %+.2e)  min %.6f @%d (gap of min %+.2e)  largest per-check gap %+.2e @%d  decays %s  exchanges %d same %s"
              % (n, kind, r, r - ref, e["min"], e["at"], e["min"] - e0["min"], worst[1], worst[0], rec[kind]["decays"], ex, same), flush=True)
        print("        per-check gaps: " + " ".join("%d:%+.1e" % g for g in gaps), flush=True)
    out["runs"][str(n)] = rec
if "out" in opts:
    with open(opts["out"], "w") as fh:
        json.dump(out, fh, indent=1)
