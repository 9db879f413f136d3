import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import test_gpu_sharded as t
for wl, f, its in (("netflix", 128, 8000),):
    ref, lr = t._unsharded(wl, f, converged_iters=its)
    e0 = dict(t._unsharded.last_extra)
    print(wl, "N=1 final %.6f lr %.2g min %.6f @%d" % (ref, lr, e0["min"], e0["at"]), [c for c in e0["checks"]], flush=True)
    for n in (8,):
        r, ex, same, lr = t._sharded_run(wl, f, n, converged_iters=its)
        e = t._sharded_run.last_extra
        print(wl, "N=%d final %.6f (gap %+.2e) lr %.2g min %.6f @%d (gap of min %+.2e) exchanges %d same %s" % (n, r, r - ref, lr, e["min"], e["at"], e["min"] - e0["min"], ex, same), [c for c in e["checks"]], flush=True)
