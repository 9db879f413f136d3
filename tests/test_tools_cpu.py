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
