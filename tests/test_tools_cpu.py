"""Host-side logic of the measurement tools (no GPU): the schedule replay of tools/tier2_converged.py against the reference's rule
(training.cu:129,146-155: patience consumed when the test RMSE got worse, never restored; decay and reset at zero)."""
import os
import sys

import numpy as np

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))


def _reference_rule(values, patience, decay, lr0):
    """training.cu:101-103,129,146-155 written out literally."""
    validation, cur, lr, decays = np.float32(np.finfo(np.float32).max), patience, np.float32(lr0), []
    for it, v in values:
        last = validation
        validation = np.float32(v)
        if last < validation:
            cur -= 1
        if cur <= 0:
            cur = patience
            lr = np.float32(lr * np.float32(decay))
            decays.append(it)
    return decays, float(lr)


def test_tier2_schedule_replay_is_the_reference_rule():
    import tier2_converged as t2
    rng = np.random.RandomState(5)
    for _ in range(50):
        n = int(rng.randint(2, 30))
        losses = np.full(500 * n, np.nan, np.float32)
        walk = np.cumsum(rng.normal(0, 0.01, n + 1)).astype(np.float32) + 0.9
        losses[0] = walk[0]
        for k in range(1, n + 1):
            losses[500 * k - 1] = walk[k]
        checks, decays, lr = t2.replay_schedule(losses, 2, 0.2, 0.01, 500)
        assert [c[0] for c in checks] == [1] + [500 * k for k in range(1, n + 1)]
        want_decays, want_lr = _reference_rule(checks, 2, 0.2, 0.01)
        assert decays == want_decays and lr == want_lr


def test_tier2_schedule_replay_known_case():
    import tier2_converged as t2
    losses = np.full(3000, np.nan, np.float32)
    for i, v in ((0, 1.0), (499, .9), (999, .8), (1499, .81), (1999, .82), (2499, .83), (2999, .84)):
        losses[i] = v
    _, decays, lr = t2.replay_schedule(losses, 2, 0.2, 0.01, 500)
    assert decays == [2000, 3000] and abs(lr - 0.01 * 0.2 * 0.2) < 1e-9


def test_scale_preflight_watchdog_takes_down_everything_it_started():
    """tools/scale_preflight.py's watchdog (the driver's bench command runs under it): at the limit the child AND what it started in
    sessions of their own -- torch.distributed.run puts its ranks in new sessions, a process-group kill of the launcher misses them --
    are killed by pid, and the call returns without waiting for a pipe that an orphan would have kept open."""
    import time
    import scale_preflight as sp
    code = ("import subprocess, sys, time\n"
            "p = subprocess.Popen([sys.executable, '-c', 'import time; time.sleep(120)'], start_new_session=True)\n"
            "print(p.pid, flush=True)\n"
            "time.sleep(120)\n")
    t0 = time.time()
    rc, out = sp.run_watched([sys.executable, "-c", code], 2)
    assert rc is None and time.time() - t0 < 30
    grandchild = int(out.split()[0])
    time.sleep(0.5)
    state = None
    try:
        with open("/proc/%d/stat" % grandchild) as fh:
            state = fh.read().rsplit(")", 1)[1].split()[0]
    except OSError:
        pass
    assert state in (None, "Z"), "the grandchild survived the watchdog: state %s" % state  # (gone, or a zombie waiting for init)
    rc, out = sp.run_watched([sys.executable, "-c", "print('done')"], 30)
    assert rc == 0 and out.strip() == "done"


def test_forced_segments_and_decay_replay_follow_the_reference_rule():
    """tests/test_gpu_sharded.py's helpers for the equal-LR-history runs: the decays replayed from logged checks are training.cu's, and
    the forced segments carry the rate that was in force UP TO each check (it changes behind the check at which the patience ran out)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import importlib
    spec = importlib.util.spec_from_file_location("sharded_helpers", os.path.join(ROOT, "tests", "test_gpu_sharded.py"))
    src = open(spec.origin).read()
    ns = {"np": np}
    start, end = src.index("def forced_segments"), src.index("def _sharded_run")
    exec(src[start:end], ns)  # (the two pure functions only: the module itself needs a GPU library at import)
    checks = [(1, 1.0), (500, .9), (1000, .8), (1500, .81), (2000, .82), (2500, .83), (3000, .84)]
    decays = ns["replay_decays"](checks)
    assert decays == _reference_rule(checks, 2, 0.2, 0.01)[0] == [2000, 3000]
    seg = ns["forced_segments"](checks, decays)
    assert [s[0] for s in seg] == [c[0] for c in checks]
    lrs = [s[1] for s in seg]
    assert lrs[:5] == [float(np.float32(0.01))] * 5 and abs(lrs[5] - 0.002) < 1e-9 and abs(lrs[6] - 0.002) < 1e-9
