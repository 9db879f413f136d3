#!/usr/bin/env python3
"""Converged user-sharded runs by the PRODUCT's driver (N ranks of cu2rec_train_sharded as threads of one process on one GPU: the
in-process world of tests/test_gpu_sharded.py) against cu2rec_train on the whole set, under the reference's LR schedule: end point,
best checkpoint, decay state -- the figures of profiles/r05_sharded_converged.txt.
usage: tools/shard_converged.py [workload factors iterations shards,...]   (default: netflix 128 8000 8)"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import test_gpu_sharded as t
_a = sys.argv[1:]
_cases = ((_a[0], int(_a[1]), int(_a[2])),) if len(_a) >= 3 else (("netflix", 128, 8000),)
_shards = tuple(int(v) for v in _a[3].split(",")) if len(_a) >= 4 else (8,)
for wl, f, its in _cases:
    ref, lr = t._unsharded(wl, f, converged_iters=its)
    e0 = dict(t._unsharded.last_extra)
    print(wl, "N=1 final %.6f lr %.2g min %.6f @%d" % (ref, lr, e0["min"], e0["at"]), [c for c in e0["checks"]], flush=True)
    for n in _shards:
        r, ex, same, lr = t._sharded_run(wl, f, n, converged_iters=its)
        e = t._sharded_run.last_extra
        print(wl, "N=%d final %.6f (gap %+.2e) lr %.2g min %.6f @%d (gap of min %+.2e) exchanges %d same %s" % (n, r, r - ref, lr, e["min"], e["at"], e["min"] - e0["min"], ex, same), [c for c in e["checks"]], flush=True)
