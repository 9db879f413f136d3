"""Pins the CPU oracle (oracle/cu2rec_oracle.c) against the reference.

* known answers restated from the reference's own tests (tests/test_util.cu, test_loss.cu,
  test_config.cu),
* stdout of the unmodified reference CPU twin (lr = 0) -- tests/golden/ref_lr0_known_answers.json,
* bit-exact P/Q/bias dumps of the reference CPU twin with the counter-based sampler --
  tests/golden/ref_sgd_golden.json (both written by oracle/gen_golden.py).
"""
import base64
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, ML_SMALL, golden_path
from oracle import oracle as orc


def _load(name):
    with open(os.path.join(GOLDEN, name)) as fh:
        return json.load(fh)


def _cfg(fields):
    names = ["cur_iterations", "total_iterations", "n_factors", "learning_rate", "seed", "P_reg", "Q_reg",
             "user_bias_reg", "item_bias_reg"]
    return orc.default_config(**dict(zip(names, fields)))


def _fmt(x):
    return "%f" % x


# ------------------------------------------------------------------ reader + CSR (tests/test_util.cu)

def test_read_csv_known_answer():
    m = orc.read_csv(os.path.join(GOLDEN, "toy_ratings.csv"))  # tests/test_util.cu:20-34
    assert (m.rows, m.cols, m.nnz) == (6, 5, 18)
    assert abs(m.global_bias - 3.5556) < 1e-3


def test_csr_known_answer():
    m = orc.read_csv(os.path.join(GOLDEN, "toy_ratings.csv"))  # tests/test_util.cu:98-142
    assert m.indptr.tolist() == [0, 4, 7, 10, 13, 16, 18]
    assert m.indices.tolist() == [0, 1, 2, 4, 0, 1, 2, 0, 1, 2, 0, 1, 2, 1, 3, 4, 3, 4]
    assert m.data.tolist() == [1, 1, 1, 5, 3, 3, 3, 4, 4, 4, 5, 5, 5, 2, 4, 4, 5, 5]


def test_csr_missing_user():
    m = orc.read_csv(os.path.join(GOLDEN, "toy_missing_user.csv"))  # tests/test_util.cu:146-189
    assert m.indptr.tolist() == [0, 4, 4, 7, 10, 13, 15]
    assert m.indices.tolist() == [0, 1, 2, 4, 0, 1, 2, 0, 1, 2, 1, 3, 4, 3, 4]
    assert m.data.tolist() == [1, 1, 1, 5, 4, 4, 4, 5, 5, 5, 2, 4, 4, 5, 5]


def test_csv_spaces_and_no_trailing_newline():
    m = orc.read_csv(os.path.join(GOLDEN, "toy_user_spaces.csv"))  # data/test/test_user_ratings.csv
    assert (m.rows, m.cols, m.nnz) == (1, 4, 3)
    assert m.indices.tolist() == [0, 1, 3] and m.data.tolist() == [1, 1, 5]


def test_config_file_format(tmp_path):
    p = tmp_path / "c.cfg"
    p.write_text("0 100 10 0.0001 42 0.2 0.1 0.1 0.1\n")  # data/test/test_config.cfg, tests/test_config.cu:11-25
    c = orc.read_config(str(p))
    assert c.total_iterations == 100 and c.n_factors == 10 and c.seed == 42
    assert abs(c.P_reg - 0.2) < 1e-7 and abs(c.learning_rate - 1e-4) < 1e-10
    q = tmp_path / "d.cfg"
    orc.lib().orc_config_write(str(q).encode(), c)  # tests/test_config.cu:27-35 round trip
    d = orc.read_config(str(q))
    for name, _ in orc.Config._fields_[:9]:
        assert getattr(c, name) == getattr(d, name)
    assert (c.n_threads, c.check_error, c.patience) == (32, 500, 2.0)  # config.h:43-48 defaults


# ------------------------------------------------------------------ init (util.cu:124-144)

def test_normal_init_known_answers():
    # values printed by the compiled reference (seed 42, stddev 1/n_factors)
    a = orc.normal_fill(6, 10)
    np.testing.assert_array_equal(a, np.array([0.122192137, -0.051696416, 0.086963594, 0.0721332654, 0.158855632,
                                               0.161821708], np.float32))
    b = orc.normal_fill(6, 1)
    np.testing.assert_array_equal(b, np.array([1.22192132, -0.516964138, 0.86963594, 0.721332669, 1.58855629,
                                               1.61821711], np.float32))
    c = orc.normal_fill(4, 100)
    np.testing.assert_array_equal(c, np.array([0.0122192129, -0.00516964123, 0.00869635958, 0.00721332664],
                                              np.float32))


# ------------------------------------------------------------------ loss (tests/test_loss.cu)

def test_loss_74():
    m = orc.read_csv(os.path.join(GOLDEN, "toy_ratings.csv"))  # tests/test_loss.cu:23-101
    f = 2
    P, Q = np.ones((m.rows, f), np.float32), np.ones((m.cols, f), np.float32)
    ub, ib = np.ones(m.rows, np.float32), np.ones(m.cols, np.float32)
    for order in (orc.DOT_SEQ, orc.DOT_TREE16):
        r = orc.loss(m, P, Q, ub, ib, 1.0, dot_order=order, want_errors=True)
        assert float(np.sum(r["errors"].astype(np.float64) ** 2)) == 74.0
        assert r["sum_sq"] == 74.0


@pytest.mark.parametrize("n", [1, 33, 1 << 10, 1 << 16])
def test_total_loss_all_ones(n):
    mae, rmse = orc.error_metrics(np.ones(n, np.float32))  # tests/test_loss.cu:106-147
    assert mae == 1.0 and rmse == 1.0


# ------------------------------------------------------------------ vs the compiled reference

def _run_case_lines(case, schedule=orc.SCHED_SEQUENTIAL):
    tr, te = orc.read_csv(golden_path(case["train"])), orc.read_csv(golden_path(case["test"]))
    cfg = _cfg(case["cfg"])
    f = cfg.n_factors
    P, Q, ub, ib = orc.init_model(tr.rows, tr.cols, f)
    log = orc.train(tr, te, cfg, P, Q, ub, ib, tr.global_bias, dot_order=orc.DOT_SEQ, acc=orc.ACC_F32,
                    schedule=schedule)
    lines = []
    for e in log:
        lines.append({"split": "TRAIN", "iteration": e["iteration"], "mae": _fmt(e["train_mae"]),
                      "rmse": _fmt(e["train_rmse"])})
        lines.append({"split": "TEST", "iteration": e["iteration"], "mae": _fmt(e["test_mae"]),
                      "rmse": _fmt(e["test_rmse"])})
    return lines, (P, Q, ub, ib)


def _needs(case):
    if "ML_SMALL" in (case["train"], case["test"]) and not os.path.exists(ML_SMALL):
        pytest.skip("reference dataset not present on this machine")


@pytest.mark.parametrize("idx", range(8))
def test_lr0_known_answers(idx):
    cases = _load("ref_lr0_known_answers.json")["cases"]
    if idx >= len(cases):
        pytest.skip("no such case")
    case = cases[idx]
    _needs(case)
    lines, _ = _run_case_lines(case)
    assert lines == case["lines"]


def test_lr0_survey_values():
    # BASELINE.md section 2 / SURVEY 8c: toy train + toy2 test, f=10
    case = [c for c in _load("ref_lr0_known_answers.json")["cases"]
            if c["train"] == "toy_ratings.csv" and c["cfg"][2] == 10][0]
    got = {(l["split"]): (l["mae"], l["rmse"]) for l in case["lines"]}
    assert got["TRAIN"] == ("1.135758", "1.426464") and got["TEST"] == ("1.233287", "1.516153")


_SGD = _load("ref_sgd_golden.json")["cases"]


@pytest.mark.parametrize("idx", range(len(_SGD)))
def test_sgd_bit_exact_vs_reference(idx):
    case = _SGD[idx]
    _needs(case)
    lines, arrays = _run_case_lines(case)
    assert lines == case["lines"]
    for name, arr in zip(("P", "Q", "user_bias", "item_bias"), arrays):
        g = case[name]
        raw = np.ascontiguousarray(arr, np.float32).tobytes()
        if "f32_le_b64" in g:
            want = np.frombuffer(base64.b64decode(g["f32_le_b64"]), np.float32)
            np.testing.assert_array_equal(arr.ravel(), want, err_msg=name)
        assert hashlib.sha256(raw).hexdigest() == g["sha256"], name


# ------------------------------------------------------------------ Tier 3: the unmodified reference binary's own band

def _tier3_runs(seeds, fields, tr, inclusive):
    L = orc.lib()
    L.orc_set_inclusive_range(1 if inclusive else 0, tr.nnz)
    try:
        out = []
        for seed in seeds:
            f2 = list(fields)
            f2[4] = seed
            cfg = _cfg(f2)
            P, Q, ub, ib = orc.init_model(tr.rows, tr.cols, cfg.n_factors)
            log = orc.train(tr, tr, cfg, P, Q, ub, ib, tr.global_bias, dot_order=orc.DOT_SEQ, acc=orc.ACC_F32, schedule=orc.SCHED_SEQUENTIAL)
            out.append({e["iteration"]: e for e in log})
        return out
    finally:
        L.orc_set_inclusive_range(0, 0)


def test_oracle_trajectory_lies_in_the_unmodified_reference_band():
    """SURVEY.md section 8c, Tier 3.  The UNMODIFIED reference CPU twin seeds a fresh std::random_device per update and draws from an
    INCLUSIVE range (mf_sequential.cu:109-112): no seed reaches its sampler, identical runs differ, and with probability 1 / (n + 1) a
    user trains on the next user's first rating.  tests/golden/ref_tier3_band.json holds min / max of its printed train RMSE / MAE
    over 8 runs on the reference's bundled ml-latest-small at f=10, lr .01, reg .02, at iterations 1 and 500 (oracle/gen_golden.py).
    (a) The oracle's arithmetic (pinned bit for bit above) driven by a sampler with the reference's DISTRIBUTION -- counter-based
        draws from the inclusive range, orc_set_inclusive_range -- follows the same trajectory: for five seeds, train RMSE and MAE
        at both iterations lie inside the band widened by 5e-4 (the reference's own run-to-run spread is +-4.5e-4).
    (b) The sampler the product uses everywhere -- the half-open range of sgd.cu:37, i.e. WITHOUT the reference's off-by-one --
        converges slightly faster: its mean train RMSE after 500 iterations sits 0.2e-3 .. 1.5e-3 BELOW the reference's mean
        (measured: 0.87283 against 0.87358; the inclusive emulation accounts for most of it: 0.87329).  Stated, not hidden."""
    if not os.path.exists(ML_SMALL):
        pytest.skip("reference dataset not present on this machine")
    gold = _load("ref_tier3_band.json")
    assert gold["runs"] >= 6
    tr = orc.read_csv(ML_SMALL)
    slack = 5e-4
    for seen in _tier3_runs((42, 7, 20240917, 3, 11), gold["cfg"], tr, inclusive=True):
        for it in (1, 500):
            band, e = gold["band"][str(it)], seen[it]
            assert band["rmse"]["min"] - slack <= e["train_rmse"] <= band["rmse"]["max"] + slack, (it, e["train_rmse"], band["rmse"])
            assert band["mae"]["min"] - slack <= e["train_mae"] <= band["mae"]["max"] + slack, (it, e["train_mae"], band["mae"])
    half_open = _tier3_runs(range(1, 9), gold["cfg"], tr, inclusive=False)
    ref_mean = float(np.mean(gold["band"]["500"]["rmse"]["runs"]))
    ours = float(np.mean([s[500]["train_rmse"] for s in half_open]))
    assert 2e-4 <= ref_mean - ours <= 1.5e-3, (ref_mean, ours)
    first = float(np.mean([s[1]["train_rmse"] for s in half_open]))
    assert abs(first - float(np.mean(gold["band"]["1"]["rmse"]["runs"]))) <= 5e-4  # one iteration in, the two samplers have not parted yet


# ------------------------------------------------------------------ internal consistency of the two dot orders

def test_tree16_order_close_to_reference_order():
    tr = orc.read_csv(os.path.join(GOLDEN, "toy_ratings3.csv"))
    rng = np.random.RandomState(0)
    for f in (1, 2, 10, 50, 64, 100, 128, 300):
        P = rng.randn(tr.rows, f).astype(np.float32)
        Q = rng.randn(tr.cols, f).astype(np.float32)
        ub, ib = rng.randn(tr.rows).astype(np.float32), rng.randn(tr.cols).astype(np.float32)
        a = orc.loss(tr, P, Q, ub, ib, 3.5, dot_order=orc.DOT_SEQ, want_errors=True)["errors"]
        b = orc.loss(tr, P, Q, ub, ib, 3.5, dot_order=orc.DOT_TREE16, want_errors=True)["errors"]
        ref = np.array([r - (3.5 + ub[u] + ib[i] + float(np.dot(Q[i].astype(np.float64), P[u].astype(np.float64))))
                        for u in range(tr.rows) for i, r in zip(tr.indices[tr.indptr[u]:tr.indptr[u + 1]],
                                                                tr.data[tr.indptr[u]:tr.indptr[u + 1]])])
        scale = 1e-6 * (1 + np.sqrt(f)) * max(1.0, np.abs(ref).max())
        assert np.abs(a - ref).max() < 10 * scale and np.abs(b - ref).max() < 10 * scale
