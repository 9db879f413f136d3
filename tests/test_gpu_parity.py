"""Parity tests proper (-m gpu): the HIP path, called through the C ABI, against the CPU oracle
on the same seeded inputs, against the committed golden vectors from the reference, and
through size-independent properties.  Bit-exact where the arithmetic order is the same
(oracle TREE16 order == the kernels' order); tolerance stated in the test otherwise.
"""
import base64
import json
import os
import subprocess

import numpy as np
import pytest

import cu2rec_amd as cu
from cu2rec_amd import synth
from conftest import GOLDEN, ROOT
from oracle import oracle as orc

pytestmark = pytest.mark.gpu

HYPER = (0.01, 0.02, 0.02, 0.02, 0.02)


def _toy(name="toy_ratings.csv"):
    return cu.createSparseMatrix(os.path.join(GOLDEN, name))


def _as_orc(m):
    return orc.CSR(m.indptr, m.indices, m.data, m.rows, m.cols, m.global_bias)


def _random_model(rows, cols, f, seed=0, scale=0.1):
    rng = np.random.RandomState(seed)
    return ((rng.randn(rows, f) * scale).astype(np.float32), (rng.randn(cols, f) * scale).astype(np.float32),
            (rng.randn(rows) * scale).astype(np.float32), (rng.randn(cols) * scale).astype(np.float32))


def _small_set(users=300, items=120, nnz=6000, seed=1):
    tr, te = synth.make_ratings(users, items, nnz, min_degree=3, seed=seed)
    return tr, te


# ------------------------------------------------------------------ loss

def test_loss_golden_74():
    m = _toy()  # tests/test_loss.cu:23-101: P=Q=1 (f=2), biases 1, global_bias forced to 1 -> sum e^2 == 74
    d = cu.DeviceCSR(m)
    model = cu.Model(m.rows, m.cols, 2, 1.0, P=np.ones((m.rows, 2)), Q=np.ones((m.cols, 2)), user_bias=np.ones(m.rows),
                     item_bias=np.ones(m.cols))
    out = model.loss(d)
    assert out["sum_sq"] == 74.0
    assert out["rmse"] == np.float32(np.sqrt(74.0 / 18))


@pytest.mark.parametrize("n", [1, 33, 1 << 10, 1 << 16, (1 << 20) + 7])
def test_total_loss_all_ones(n):
    from cu2rec_amd.engine import Engine
    eng = Engine(1, 1, 4, 0.0)  # tests/test_loss.cu:106-147: all-ones residuals -> mae == rmse == 1 exactly
    mae, rmse = eng.error_metrics(np.ones(n, np.float32))
    assert mae == 1.0 and rmse == 1.0


@pytest.mark.parametrize("f", [1, 2, 10, 50, 64, 100, 128, 300])
def test_loss_bit_exact_vs_oracle(f):
    from cu2rec_amd.engine import DeviceRatings, Engine
    tr, _ = _small_set()
    P, Q, ub, ib = _random_model(tr.rows, tr.cols, f, seed=f)
    eng = Engine(tr.rows, tr.cols, f, tr.global_bias, P, Q, ub, ib)
    got = eng.loss(DeviceRatings(tr, eng.device), want_errors=True)
    want = orc.loss(_as_orc(tr), P, Q, ub, ib, tr.global_bias, dot_order=orc.DOT_TREE16, acc=orc.ACC_F64,
                    want_errors=True)
    np.testing.assert_array_equal(got["errors"], want["errors"])  # residuals: same order of operations -> same bits
    # the double sums are added in a different order (per group, per block, then blocks): 1e-12 relative
    assert abs(got["sum_sq"] - want["sum_sq"]) <= 1e-12 * want["sum_sq"]
    assert abs(got["sum_abs"] - want["sum_abs"]) <= 1e-12 * want["sum_abs"]
    assert got["rmse"] == want["rmse"] and got["mae"] == want["mae"]
    # against the reference's own summation order (sequential dot): float rounding of the dot only
    ref = orc.loss(_as_orc(tr), P, Q, ub, ib, tr.global_bias, dot_order=orc.DOT_SEQ, acc=orc.ACC_F64)
    assert abs(got["rmse"] - ref["rmse"]) <= 2e-6 * max(1.0, ref["rmse"])


def test_loss_empty_users_and_ragged_rows():
    m = _toy("toy_missing_user.csv")  # user 2 has no ratings
    P, Q, ub, ib = _random_model(m.rows, m.cols, 10, seed=3, scale=1.0)
    model = cu.Model(m.rows, m.cols, 10, m.global_bias, P, Q, ub, ib)
    got = model.loss(cu.DeviceCSR(m))
    want = orc.loss(_as_orc(m), P, Q, ub, ib, m.global_bias, dot_order=orc.DOT_TREE16)
    assert got["rmse"] == want["rmse"] and got["mae"] == want["mae"]
    # a test set with fewer users than the model (Appendix quirk 6)
    t = _toy("toy_user_spaces.csv")
    got = model.loss(cu.DeviceCSR(cu.HostCSR(t.indptr, t.indices, t.data, t.rows, m.cols, t.global_bias)))
    want = orc.loss(_as_orc(t), P, Q, ub, ib, m.global_bias, dot_order=orc.DOT_TREE16)
    assert got["rmse"] == want["rmse"]


# ------------------------------------------------------------------ SGD, serial order: exact

@pytest.mark.parametrize("f,iters", [(1, 7), (2, 10), (10, 25), (50, 10), (100, 10), (128, 5), (300, 3)])
def test_sgd_serial_bit_exact_vs_oracle(f, iters):
    tr, _ = _small_set()
    P, Q, ub, ib = orc.init_model(tr.rows, tr.cols, f)
    model = cu.Model(tr.rows, tr.cols, f, tr.global_bias)  # library-side seed-42 init
    np.testing.assert_array_equal(model.download()[0], P)
    model.sgd(cu.DeviceCSR(tr), HYPER, seed=42, iter0=0, n_iters=iters, mode="serial")
    orc.sgd_iterations(_as_orc(tr), P, Q, ub, ib, tr.global_bias, HYPER, 42, 0, iters, dot_order=orc.DOT_TREE16)
    gP, gQ, gub, gib = model.download()
    np.testing.assert_array_equal(gP, P)
    np.testing.assert_array_equal(gQ, Q)
    np.testing.assert_array_equal(gub, ub)
    np.testing.assert_array_equal(gib, ib)


def test_sgd_serial_resume_and_frozen_items():
    tr, _ = _small_set(seed=4)
    f = 10
    P, Q, ub, ib = orc.init_model(tr.rows, tr.cols, f)
    model = cu.Model(tr.rows, tr.cols, f, tr.global_bias)
    d = cu.DeviceCSR(tr)
    model.sgd(d, HYPER, 7, 0, 3, mode="serial")
    model.sgd(d, HYPER, 7, 3, 4, mode="serial")  # resumed stream == one run of 7
    orc.sgd_iterations(_as_orc(tr), P, Q, ub, ib, tr.global_bias, HYPER, 7, 0, 7, dot_order=orc.DOT_TREE16)
    for g, w in zip(model.download(), (P, Q, ub, ib)):
        np.testing.assert_array_equal(g, w)
    # is_train == false: Q and item_bias stay put (config.h:41, sgd.cu:61,70)
    model.sgd(d, HYPER, 7, 7, 2, mode="serial", update_items=False)
    Q0, ib0 = Q.copy(), ib.copy()
    orc.sgd_iterations(_as_orc(tr), P, Q, ub, ib, tr.global_bias, HYPER, 7, 7, 2, dot_order=orc.DOT_TREE16,
                       update_items=False)
    gP, gQ, gub, gib = model.download()
    np.testing.assert_array_equal(gQ, Q0)
    np.testing.assert_array_equal(gib, ib0)
    np.testing.assert_array_equal(gP, P)


def test_sgd_serial_vs_reference_golden():
    """GPU (serial order) against the dumps of the reference CPU twin itself.  The only difference
    is the summation order of the f-term dot product (tree vs sequential): <= 1e-5 absolute on every
    parameter for these runs (north star: 1e-4 RMSE)."""
    cases = json.load(open(os.path.join(GOLDEN, "ref_sgd_golden.json")))["cases"]
    n = 0
    for case in cases:
        if "f32_le_b64" not in case["P"]:
            continue  # the reference's bundled dataset does not travel
        tr = _toy(case["train"])
        cur, total, f, lr, seed, pr, qr, ur, ir = case["cfg"]
        model = cu.Model(tr.rows, tr.cols, f, tr.global_bias)
        model.sgd(cu.DeviceCSR(tr), (lr, pr, qr, ur, ir), seed, cur, total, mode="serial")
        got = model.download()
        for name, g in zip(("P", "Q", "user_bias", "item_bias"), got):
            want = np.frombuffer(base64.b64decode(case[name]["f32_le_b64"]), np.float32).reshape(g.shape)
            assert np.abs(g - want).max() <= 1e-5, (case["cfg"], name)
        te = _toy(case["test"])
        rmse = model.loss(cu.DeviceCSR(cu.HostCSR(te.indptr, te.indices, te.data, te.rows, tr.cols)))["rmse"]
        assert abs(rmse - float(case["lines"][-1]["rmse"])) <= 1e-5
        n += 1
    assert n >= 10


# ------------------------------------------------------------------ SGD, Hogwild

def test_sgd_hogwild_exact_when_no_item_is_shared():
    # every user rates only its own item: no two updates of an iteration touch the same row, so
    # the parallel schedule has the sequential one's result bit for bit
    n, f = 5000, 100
    rng = np.random.RandomState(0)
    m = cu.HostCSR(np.arange(n + 1), rng.permutation(n), rng.randint(1, 6, n).astype(np.float32), n, n, 3.0)
    P, Q, ub, ib = orc.init_model(n, n, f)
    model = cu.Model(n, n, f, 3.0)
    model.sgd(cu.DeviceCSR(m), HYPER, 42, 0, 20, mode="hogwild")
    orc.sgd_iterations(_as_orc(m), P, Q, ub, ib, 3.0, HYPER, 42, 0, 20, dot_order=orc.DOT_TREE16)
    for g, w in zip(model.download(), (P, Q, ub, ib)):
        np.testing.assert_array_equal(g, w)


def test_factor_range_limits():
    """n_factors up to 512 is compiled in (J = 8 slots per lane); beyond that the library refuses, loudly."""
    tr, _ = _small_set(users=60, items=40, nnz=600, seed=13)
    f = 512
    P, Q, ub, ib = orc.init_model(tr.rows, tr.cols, f)
    model = cu.Model(tr.rows, tr.cols, f, tr.global_bias)
    d = cu.DeviceCSR(tr)
    model.sgd(d, HYPER, 42, 0, 3, mode="ordered")
    orc.sgd_iterations(_as_orc(tr), P, Q, ub, ib, tr.global_bias, HYPER, 42, 0, 3, dot_order=orc.DOT_TREE16)
    for g, w in zip(model.download(), (P, Q, ub, ib)):
        np.testing.assert_array_equal(g, w)
    assert model.loss(d)["rmse"] == orc.loss(_as_orc(tr), P, Q, ub, ib, tr.global_bias, dot_order=orc.DOT_TREE16)["rmse"]
    with pytest.raises(cu.Cu2recError) as e:
        cu.Model(tr.rows, tr.cols, 513, tr.global_bias)
    assert e.value.status == -6  # CU2REC_EUNSUPPORTED


@pytest.mark.parametrize("f", [10, 100, 300])
def test_sgd_hogwild_one_iteration_is_jacobi(f):
    """One Hogwild iteration: every user's P row / bias is the update computed from the item row as it
    was at the start of the launch or as some concurrent update left it; for items sampled by exactly
    one user this is the start-of-iteration row, so those users and items must match the oracle's
    single update bit for bit.  Shared items must hold values produced by one of their updates."""
    tr, _ = _small_set(users=2000, items=300, nnz=40000, seed=9)
    P0, Q0, ub0, ib0 = _random_model(tr.rows, tr.cols, f, seed=2)
    model = cu.Model(tr.rows, tr.cols, f, tr.global_bias, P0, Q0, ub0, ib0)
    model.sgd(cu.DeviceCSR(tr), HYPER, 42, 5, 1, mode="hogwild")
    gP, gQ, gub, gib = model.download()
    items = np.array([tr.indices[orc.sample(42, u, 5, tr.indptr[u], tr.indptr[u + 1])] for u in range(tr.rows)])
    counts = np.bincount(items, minlength=tr.cols)
    o = _as_orc(tr)
    n_unique = 0
    for u in range(tr.rows):
        if counts[items[u]] != 1:
            continue
        P, Q, ub, ib = P0.copy(), Q0.copy(), ub0.copy(), ib0.copy()
        orc.sgd_one(o, u, P, Q, ub, ib, tr.global_bias, HYPER, 42, 5, dot_order=orc.DOT_TREE16)
        np.testing.assert_array_equal(gP[u], P[u])
        np.testing.assert_array_equal(gQ[items[u]], Q[items[u]])
        assert gub[u] == ub[u] and gib[items[u]] == ib[items[u]]
        n_unique += 1
    assert n_unique > 20
    untouched = counts == 0
    np.testing.assert_array_equal(gQ[untouched], Q0[untouched])
    assert np.isfinite(gP).all() and np.isfinite(gQ).all()  # tests/test_sgd.cu:134-145


def test_sgd_hogwild_converges_like_the_oracle():
    """Hogwild vs the sequential oracle on the same sample stream: test RMSE after 300 iterations within
    2e-2 (lost updates on popular items slow Hogwild down a little), both well below the
    starting RMSE."""
    tr, te = _small_set(users=3000, items=400, nnz=90000, seed=11)
    f, iters = 16, 300
    P, Q, ub, ib = orc.init_model(tr.rows, tr.cols, f)
    start = orc.loss(_as_orc(te), P, Q, ub, ib, tr.global_bias)["rmse"]
    orc.sgd_iterations(_as_orc(tr), P, Q, ub, ib, tr.global_bias, HYPER, 42, 0, iters, dot_order=orc.DOT_TREE16)
    want = orc.loss(_as_orc(te), P, Q, ub, ib, tr.global_bias)["rmse"]
    model = cu.Model(tr.rows, tr.cols, f, tr.global_bias)
    model.sgd(cu.DeviceCSR(tr), HYPER, 42, 0, iters, mode="hogwild")
    got = model.loss(cu.DeviceCSR(te))["rmse"]
    assert want < start - 0.05 and got < start - 0.05
    assert abs(got - want) < 2e-2


# ------------------------------------------------------------------ train() and the CLI

def test_train_schedule_matches_oracle():
    tr, te = _small_set(seed=5)
    cfg = cu.default_config(total_iterations=40, n_factors=10, check_error=8, learning_rate=0.05, patience=1.0)
    ocfg = orc.default_config(total_iterations=40, n_factors=10, check_error=8, learning_rate=0.05, patience=1.0)
    P, Q, ub, ib = orc.init_model(tr.rows, tr.cols, 10)
    log = orc.train(_as_orc(tr), _as_orc(te), ocfg, P, Q, ub, ib, tr.global_bias, dot_order=orc.DOT_TREE16,
                    acc=orc.ACC_F64, schedule=orc.SCHED_PATIENCE)
    gP, gQ, losses, gub, gib, stats = cu.train(tr, te, cfg, mode="serial", verbose=False, return_stats=True)
    np.testing.assert_array_equal(gP, P)
    np.testing.assert_array_equal(gQ, Q)
    checked = [e["iteration"] - 1 for e in log]
    assert checked == [0, 7, 15, 23, 31, 39]  # training.cu:118
    for e in log:
        assert losses[e["iteration"] - 1] == e["test_rmse"]
    assert np.isnan(losses[[i for i in range(40) if i not in checked]]).all()
    assert cfg.learning_rate == ocfg.learning_rate and cfg.cur_iterations == 40 == ocfg.cur_iterations
    assert stats.n_checks == 6 and stats.last_test_rmse == log[-1]["test_rmse"]


def test_training_loop_loss_goes_down():
    m = _toy()  # tests/test_training.cu:20-55: 10 iterations, f=2, lr 1e-3, reg .1, train == test
    cfg = cu.default_config(total_iterations=10, n_factors=2, learning_rate=1e-3, P_reg=0.1, Q_reg=0.1,
                            user_bias_reg=0.1, item_bias_reg=0.1, seed=42)
    _, _, losses, _, _ = cu.train(m, m, cfg, verbose=False)
    assert losses[0] >= losses[9]


def test_bin_mf_cli(tmp_path):
    exe = os.path.join(ROOT, "bin", "mf")
    assert os.path.exists(exe), "bin/mf not built"
    train = tmp_path / "ratings.csv"
    train.write_text(open(os.path.join(GOLDEN, "toy_ratings.csv")).read())
    cfgp = tmp_path / "c.cfg"
    cfgp.write_text("0 20 4 0.01 42 0.02 0.02 0.02 0.02\n")
    out = subprocess.run([exe, "-c", str(cfgp), "-m", "serial", str(train), os.path.join(GOLDEN, "toy_ratings2.csv")],
                         stdout=subprocess.PIPE, text=True, check=True).stdout
    assert "Free memory:" in out and "Hyperparameters:" in out and "n_factors: 4" in out
    assert "TRAIN: Iteration 1 GPU MAE:" in out and "TEST: Iteration 20 GPU MAE:" in out
    assert "Time taken for 20 of iterations is" in out
    # five component files next to the training file (mf.cu:83-87)
    tr = _toy()
    P, Q, ub, ib = orc.init_model(tr.rows, tr.cols, 4)
    orc.sgd_iterations(_as_orc(tr), P, Q, ub, ib, tr.global_bias, HYPER, 42, 0, 20, dot_order=orc.DOT_TREE16)
    for comp, want in (("p", P), ("q", Q), ("user_bias", ub), ("item_bias", ib)):
        path = tmp_path / ("ratings_f4_%s.csv" % comp)
        orc.write_csv(str(tmp_path / "want.csv"), want)
        assert path.read_text() == (tmp_path / "want.csv").read_text(), comp
    assert (tmp_path / "ratings_f4_global_bias.csv").read_text() == "%f\n" % tr.global_bias
    assert subprocess.run([exe]).returncode == 255  # mf.cu:17-19: return -1
    assert subprocess.run([exe, "-z"], stdout=subprocess.PIPE, stderr=subprocess.PIPE).returncode == 1


# ------------------------------------------------------------------ larger shapes: properties

def test_ml1m_shape_properties():
    """ML-1M shape, f=50 (BASELINE.json configs[1]): loss agrees with the oracle to 1e-6, RMSE falls
    monotonically over the first checks, parameters stay finite, untouched items keep their rows."""
    tr, te = synth.make_named("ml-1m")
    f = 50
    model = cu.Model(tr.rows, tr.cols, f, tr.global_bias)
    dtr, dte = cu.DeviceCSR(tr), cu.DeviceCSR(te)
    P, Q, ub, ib = orc.init_model(tr.rows, tr.cols, f)
    want = orc.loss(_as_orc(te), P, Q, ub, ib, tr.global_bias, dot_order=orc.DOT_TREE16)
    got = model.loss(dte)
    assert got["rmse"] == want["rmse"] and abs(got["sum_sq"] - want["sum_sq"]) <= 1e-12 * want["sum_sq"]
    last = got["rmse"]
    for k in range(4):
        model.sgd(dtr, HYPER, 42, 100 * k, 100, mode="hogwild")
        cur = model.loss(dte)["rmse"]
        assert cur < last
        last = cur
    for a in model.download():
        assert np.isfinite(a).all()


# ------------------------------------------------------------------ SGD, ordered: the sequential result, in parallel

@pytest.mark.parametrize("f,iters", [(1, 5), (10, 150), (50, 20), (100, 70), (128, 5), (300, 3)])
def test_sgd_ordered_bit_exact_vs_oracle(f, iters):
    # 300 users x 120 items: every iteration has long item chains; 150 and 70 cross the 64-iteration batches
    tr, _ = _small_set()
    P, Q, ub, ib = orc.init_model(tr.rows, tr.cols, f)
    model = cu.Model(tr.rows, tr.cols, f, tr.global_bias)
    model.sgd(cu.DeviceCSR(tr), HYPER, seed=42, iter0=0, n_iters=iters, mode="ordered")
    orc.sgd_iterations(_as_orc(tr), P, Q, ub, ib, tr.global_bias, HYPER, 42, 0, iters, dot_order=orc.DOT_TREE16)
    for name, g, w in zip("P Q ub ib".split(), model.download(), (P, Q, ub, ib)):
        np.testing.assert_array_equal(g, w, err_msg=name)


def test_sgd_ordered_schedule_with_empty_users_and_rows_above_2048_ratings():
    """The schedule's corner cases in one set: users without ratings inside the range, a rating row above 2,048 ratings (the
    row-reading key kernel's gather fallback) and 70 iterations (two windows).  The ordered mode against the oracle bit for bit, and
    the block-solve mode -- whose plan is built from the same sorted arrays -- within 2e-6."""
    tr, _ = synth.make_ratings(700, 3000, 60000, min_degree=3, seed=9)
    indptr = tr.indptr.copy()                       # users 100..139 lose their ratings
    lo, hi = indptr[100], indptr[140]
    indptr[100:141] = lo
    indptr[141:] -= hi - lo
    indices, data = np.delete(tr.indices, np.s_[lo:hi]), np.delete(tr.data, np.s_[lo:hi])
    rng = np.random.RandomState(3)                  # user 7 rates 2,500 items: above the row-reading kernel's 2,048
    l7, h7 = indptr[7], indptr[8]
    big_i = np.arange(2500, dtype=indices.dtype)
    big_r = (rng.randint(1, 11, size=2500) * 0.5).astype(data.dtype)
    indices = np.concatenate([indices[:l7], big_i, indices[h7:]])
    data = np.concatenate([data[:l7], big_r, data[h7:]])
    indptr[8:] += 2500 - (h7 - l7)
    tr = cu.HostCSR(indptr, indices, data, tr.rows, tr.cols, float(data.mean()))
    assert (np.diff(tr.indptr) > 2048).any() and (np.diff(tr.indptr) == 0).any()
    f, iters = 20, 70
    state = orc.init_model(tr.rows, tr.cols, f)
    orc.sgd_iterations(_as_orc(tr), *state, tr.global_bias, HYPER, 42, 0, iters, dot_order=orc.DOT_TREE16)
    m = cu.Model(tr.rows, tr.cols, f, tr.global_bias)
    m.sgd(cu.DeviceCSR(tr), HYPER, 42, 0, iters, mode="ordered")
    for g, w in zip(m.download(), state):
        np.testing.assert_array_equal(g, w)
    prev = cu.api.blocksolve_min_rate(2.0)
    try:
        d = cu.DeviceCSR(tr)
        assert d.blocksolve_items() > 0
        m = cu.Model(tr.rows, tr.cols, f, tr.global_bias)
        m.sgd(d, HYPER, 42, 0, iters, mode="blocksolve")
    finally:
        cu.api.blocksolve_min_rate(prev if prev > 0 else -1.0)
    assert max(float(np.abs(g.astype(np.float64) - w).max()) for g, w in zip(m.download(), state)) <= 2e-6


def test_sgd_ordered_calls_of_any_length_share_schedule_windows():
    """Round 4: the schedule lives in windows of max_batch (64) iterations that outlive the call -- a call runs its iterations out
    of whichever window holds them, at whatever offset.  Calls that continue each other across window borders, a jump back into a
    window already run, a jump ahead, another seed, a second model on the same ratings, block-solve calls interleaved (their windows
    carry a plan and are not the ordered mode's): each call is the oracle's bit for bit in the ordered mode."""
    tr, _ = _small_set()
    f = 10
    d = cu.DeviceCSR(tr)
    calls = [(42, 0, 7), (42, 7, 20), (42, 27, 50), (42, 77, 3), (42, 80, 70), (42, 5, 10), (42, 15, 4), (42, 400, 9), (7, 409, 30),
             (7, 439, 140)]
    P, Q, ub, ib = orc.init_model(tr.rows, tr.cols, f)
    model = cu.Model(tr.rows, tr.cols, f, tr.global_bias)
    other = cu.Model(tr.rows, tr.cols, f, tr.global_bias)
    for k, (seed, it0, n) in enumerate(calls):
        model.sgd(d, HYPER, seed, it0, n, mode="ordered")
        if k % 3 == 1:
            other.sgd(d, HYPER, seed, it0 + n, 5, mode="blocksolve")  # (another mode's window takes a slot in between)
        orc.sgd_iterations(_as_orc(tr), P, Q, ub, ib, tr.global_bias, HYPER, seed, it0, n, dot_order=orc.DOT_TREE16)
        for name, g, w in zip("P Q ub ib".split(), model.download(), (P, Q, ub, ib)):
            np.testing.assert_array_equal(g, w, err_msg="%s after call %d" % (name, k))


def test_sgd_ordered_resume_empty_users_and_frozen_items():
    m = _toy("toy_missing_user.csv")  # user 2 has no ratings: sentinel keys in the schedule
    f = 10
    P, Q, ub, ib = orc.init_model(m.rows, m.cols, f)
    model = cu.Model(m.rows, m.cols, f, m.global_bias)
    d = cu.DeviceCSR(m)
    model.sgd(d, HYPER, 9, 0, 3, mode="ordered")
    model.sgd(d, HYPER, 9, 3, 70, mode="ordered")
    orc.sgd_iterations(_as_orc(m), P, Q, ub, ib, m.global_bias, HYPER, 9, 0, 73, dot_order=orc.DOT_TREE16)
    for g, w in zip(model.download(), (P, Q, ub, ib)):
        np.testing.assert_array_equal(g, w)
    model.sgd(d, HYPER, 9, 73, 5, mode="ordered", update_items=False)
    orc.sgd_iterations(_as_orc(m), P, Q, ub, ib, m.global_bias, HYPER, 9, 73, 5, dot_order=orc.DOT_TREE16,
                       update_items=False)
    for g, w in zip(model.download(), (P, Q, ub, ib)):
        np.testing.assert_array_equal(g, w)


def test_sgd_ordered_engine_user_offset():
    """Raw-pointer entry point with a sharded user range: draws are keyed by the global user id."""
    from cu2rec_amd.engine import DeviceRatings, Engine
    tr, _ = _small_set(seed=8)
    f, u0, u1 = 20, 100, 260
    P, Q, ub, ib = orc.init_model(tr.rows, tr.cols, f)
    shard = tr.slice_users(u0, u1)
    eng = Engine(u1 - u0, tr.cols, f, tr.global_bias, P[u0:u1], Q, ub[u0:u1], ib)
    eng.sgd(DeviceRatings(shard, eng.device), HYPER, 42, 0, 12, mode="ordered", user_offset=u0)
    # oracle: the same users inside the full matrix, everyone else without ratings
    indptr = tr.indptr.copy()
    indptr[:u0 + 1] = tr.indptr[u0]
    indptr[u1:] = tr.indptr[u1]
    masked = orc.CSR(indptr, tr.indices, tr.data, tr.rows, tr.cols)
    orc.sgd_iterations(masked, P, Q, ub, ib, tr.global_bias, HYPER, 42, 0, 12, dot_order=orc.DOT_TREE16)
    gP, gQ, gub, gib = eng.download()
    np.testing.assert_array_equal(gP, P[u0:u1])
    np.testing.assert_array_equal(gQ, Q)
    np.testing.assert_array_equal(gub, ub[u0:u1])
    np.testing.assert_array_equal(gib, ib)


def test_ordered_full_size_ml1m_and_ml20m_bit_exact():
    """BASELINE.json configs[1] and [2] at full size: the ordered GPU schedule against the sequential CPU
    oracle, every parameter bit for bit (ML-1M shape f=50, 100 iterations; ML-20M shape f=100, 8 iterations)."""
    import bench
    for name, f, iters in (("ml-1m", 50, 100), ("ml-20m", 100, 8)):
        tr, te = bench.load_dataset(name, 20240917, 0, lambda: None)
        P, Q, ub, ib = orc.init_model(tr.rows, tr.cols, f)
        model = cu.Model(tr.rows, tr.cols, f, tr.global_bias)
        model.sgd(cu.DeviceCSR(tr), HYPER, 42, 0, iters, mode="ordered")
        orc.sgd_iterations(_as_orc(tr), P, Q, ub, ib, tr.global_bias, HYPER, 42, 0, iters, dot_order=orc.DOT_TREE16)
        gP, gQ, gub, gib = model.download()
        np.testing.assert_array_equal(gQ, Q, err_msg=name)
        np.testing.assert_array_equal(gP, P, err_msg=name)
        np.testing.assert_array_equal(gub, ub, err_msg=name)
        np.testing.assert_array_equal(gib, ib, err_msg=name)
        got = model.loss(cu.DeviceCSR(te))
        want = orc.loss(_as_orc(te), P, Q, ub, ib, tr.global_bias, dot_order=orc.DOT_TREE16)
        assert got["rmse"] == want["rmse"] and got["mae"] == want["mae"]


def test_train_ordered_equals_serial():
    tr, te = _small_set(seed=6)
    outs = []
    for mode in ("serial", "ordered"):
        cfg = cu.default_config(total_iterations=30, n_factors=12, check_error=10, learning_rate=0.02)
        outs.append(cu.train(tr, te, cfg, mode=mode, verbose=False))
    for a, b in zip(outs[0], outs[1]):
        np.testing.assert_array_equal(a, b)


# ------------------------------------------------------------------ multi-GPU plumbing on one GPU

def test_item_exchange_kernels_and_rccl_world1():
    """The exchange path on real hardware with world_size 1: RCCL all-reduce (backend nccl) of the fused delta
    buffer, pack / apply kernels; ShardedSGD at N=1 must be bit-identical to the plain path (no exchange)."""
    import torch
    import torch.distributed as dist
    from cu2rec_amd.engine import DeviceRatings, Engine
    from exchange_reference import ShardedSGD
    tr, te = _small_set(users=1500, items=200, nnz=30000, seed=12)
    f = 20
    if not dist.is_initialized():
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % (29500 + os.getpid() % 500), rank=0,
                                world_size=1, device_id=torch.device("cuda", 0))
    try:
        eng = Engine(tr.rows, tr.cols, f, tr.global_bias)
        ref = Engine(tr.rows, tr.cols, f, tr.global_bias)
        d = DeviceRatings(tr, eng.device)
        job = ShardedSGD(eng, d, sync_every=3)
        job.run(HYPER, 42, 0, 10, "ordered")
        ref.sgd(d, HYPER, 42, 0, 10, "ordered")
        for a, b in zip(eng.download(), ref.download()):
            np.testing.assert_array_equal(a, b)
        # manual exchange: snapshot, train, pack -> all_reduce -> apply(1.0) reproduces Q up to one rounding
        eng.snapshot_items()
        Q0 = eng.Q.clone()
        eng.sgd(d, HYPER, 42, 10, 5, "hogwild")
        Q1, ib1 = eng.Q.clone(), eng.item_bias.clone()
        buf = eng.pack_item_delta()
        torch.testing.assert_close(buf[:Q1.numel()].view_as(Q1), Q1 - Q0, rtol=0, atol=0)
        dist.all_reduce(buf)
        eng.apply_item_delta(1.0)
        torch.testing.assert_close(eng.Q, Q1, rtol=0, atol=1e-6)
        torch.testing.assert_close(eng.item_bias, ib1, rtol=0, atol=1e-6)
        torch.testing.assert_close(eng.Q_base, eng.Q, rtol=0, atol=0)
        # mean merge with scale 1/2 halves the step
        eng.sgd(d, HYPER, 42, 15, 5, "hogwild")
        Q2 = eng.Q.clone()
        eng.pack_item_delta()
        eng.apply_item_delta(0.5)
        torch.testing.assert_close(eng.Q, eng.Q_base)
        torch.testing.assert_close(eng.Q, Q1 + 0.5 * (Q2 - Q1), rtol=0, atol=1e-6)
        out = job.loss(DeviceRatings(te, eng.device))
        assert out["n"] == te.nnz and np.isfinite(out["rmse"])
        # overlapped form: async RCCL all-reduce in flight while training continues, folded in afterwards
        eng.snapshot_items()
        base = eng.Q.clone()
        eng.sgd(d, HYPER, 42, 20, 5, "hogwild")
        buf = eng.pack_item_delta()
        eng.snapshot_for_overlap()
        snap = eng.Q.clone()
        work = dist.all_reduce(buf, async_op=True)
        eng.sgd(d, HYPER, 42, 25, 5, "hogwild")  # runs next to the collective
        now = eng.Q.clone()
        work.wait()
        eng.apply_item_delta_overlapped(1.0)
        torch.testing.assert_close(eng.Q_base, snap, rtol=0, atol=1e-6)          # merged = base + (snap - base)
        torch.testing.assert_close(eng.Q, now, rtol=0, atol=2e-6)                # + local progress since the pack
        assert not torch.equal(now, snap) and not torch.equal(snap, base)
    finally:
        dist.destroy_process_group()


# ------------------------------------------------------------------ next row: bin/predict

def test_bin_predict_partial_fit(tmp_path):
    """predict.cu:72-146: load Q / item_bias / global_bias written by bin/mf, fit ONE new user with the item side
    frozen, score every item, list the unrated ones best first.  Checked against the oracle (frozen-item SGD is
    race free, so the result is exact)."""
    mf, predict = os.path.join(ROOT, "bin", "mf"), os.path.join(ROOT, "bin", "predict")
    assert os.path.exists(predict), "bin/predict not built"
    train = tmp_path / "ratings.csv"
    train.write_text(open(os.path.join(GOLDEN, "toy_ratings.csv")).read())
    (tmp_path / "train.cfg").write_text("0 50 4 0.05 42 0.02 0.02 0.02 0.02\n")
    subprocess.run([mf, "-c", str(tmp_path / "train.cfg"), "-m", "ordered", str(train),
                    os.path.join(GOLDEN, "toy_ratings2.csv")], stdout=subprocess.PIPE, check=True)
    (tmp_path / "predict.cfg").write_text("0 200 4 0.05 42 0.02 0.02 0.02 0.02\n")
    user_file = os.path.join(GOLDEN, "toy_user_spaces.csv")  # one user: items 1, 2, 4 rated
    out = subprocess.run([predict, "-c", str(tmp_path / "predict.cfg"), "-i", str(tmp_path / "ratings_f4_item_bias.csv"),
                          "-g", str(tmp_path / "ratings_f4_global_bias.csv"), "-q", str(tmp_path / "ratings_f4_q.csv"),
                          user_file], stdout=subprocess.PIPE, text=True, check=True).stdout
    assert "Predictions: " in out and "Recommendations:" in out
    # oracle: the same partial fit
    Q = cu.read_array(str(tmp_path / "ratings_f4_q.csv"))
    ib = cu.read_array(str(tmp_path / "ratings_f4_item_bias.csv")).ravel()
    gb = float(cu.read_array(str(tmp_path / "ratings_f4_global_bias.csv")).ravel()[0])
    u = _toy("toy_user_spaces.csv")
    one = orc.CSR(np.array([0, u.nnz]), u.indices, u.data, 1, Q.shape[0])
    P, ub = orc.normal_fill(4, 4).reshape(1, 4), orc.normal_fill(1, 4)
    Qc, ibc = Q.copy(), ib.copy()
    orc.sgd_iterations(one, P, Qc, ub, ibc, gb, (0.05, 0.02, 0.02, 0.02, 0.02), 42, 0, 200, dot_order=orc.DOT_TREE16,
                       update_items=False)
    np.testing.assert_array_equal(Qc, Q)
    pred = [orc.lib().orc_predict(4, P[0].ctypes.data_as(orc.C.POINTER(orc.C.c_float)),
                                  Q[i].ctypes.data_as(orc.C.POINTER(orc.C.c_float)), float(ub[0]), float(ib[i]), gb,
                                  orc.DOT_SEQ) for i in range(Q.shape[0])]
    want = sorted([(p, i) for i, p in enumerate(pred) if i not in set(u.indices.tolist())], key=lambda t: -t[0])
    lines = [l for l in out.split("\n") if l.startswith("Rank:")]
    assert len(lines) == len(want) == 2
    for rank, (line, (p, i)) in enumerate(zip(lines, want)):
        assert line == "Rank: %d\tItem: %d\tEstimated rating: %f" % (rank + 1, i, p)
    assert subprocess.run([predict]).returncode == 2  # predict.cu:73-75


def test_train_sharded_world1_equals_cpp_train(tmp_path):
    """The product's multi-GPU train driver (cu2rec_train_sharded, csrc/sharded.cpp, through cu2rec_amd.sharded) at N=1 is
    bit-identical to cu2rec_train, and the multi-GPU CLI module writes the same five files as bin/mf."""
    from cu2rec_amd.sharded import Comm, train_sharded
    tr, te = _small_set(seed=7)
    kw = dict(total_iterations=24, n_factors=12, check_error=6, learning_rate=0.03, patience=1.0)
    cfg_a, cfg_b = cu.default_config(**kw), cu.default_config(**kw)
    a = cu.train(tr, te, cfg_a, mode="ordered", verbose=False)
    b = train_sharded(Comm(0, 1), tr, te, cfg_b, mode="ordered", verbose=False)
    for x, y in zip(a, b[:5]):
        np.testing.assert_array_equal(x, y)
    assert cfg_a.learning_rate == cfg_b.learning_rate and cfg_a.cur_iterations == cfg_b.cur_iterations == 24
    # CLI: python -m cu2rec_amd.mf_mgpu (one process) vs bin/mf
    from cu2rec_amd import synth as _synth
    d1, d2 = tmp_path / "a", tmp_path / "b"
    d1.mkdir(), d2.mkdir()
    for d in (d1, d2):
        _synth.write_csv(str(d / "train.csv"), tr)
        _synth.write_csv(str(d / "test.csv"), te)
        (d / "c.cfg").write_text("0 24 12 0.03 42 0.02 0.02 0.02 0.02\n")
    env = dict(os.environ, PYTHONPATH=ROOT)
    subprocess.run([os.path.join(ROOT, "bin", "mf"), "-c", str(d1 / "c.cfg"), "-m", "ordered", str(d1 / "train.csv"),
                    str(d1 / "test.csv")], stdout=subprocess.PIPE, check=True)
    import sys
    out = subprocess.run([sys.executable, "-m", "cu2rec_amd.mf_mgpu", "-c", str(d2 / "c.cfg"), "-m", "ordered",
                          str(d2 / "train.csv"), str(d2 / "test.csv")], stdout=subprocess.PIPE, text=True, check=True,
                         env=env, cwd=ROOT).stdout
    assert "TEST: Iteration 24 GPU MAE:" in out
    for comp in ("p", "q", "user_bias", "item_bias", "global_bias"):
        assert (d1 / ("train_f12_%s.csv" % comp)).read_text() == (d2 / ("train_f12_%s.csv" % comp)).read_text(), comp


def test_converged_run_matches_mf_sequential_within_1e4():
    """North-star criterion on BASELINE.json configs[1] (ML-1M shape, f=50): a full train() run -- 2,000 iterations,
    loss every 500, patience / LR decay -- in ordered mode against (a) the oracle in the kernels' summation order:
    every parameter and every logged loss bit for bit, and (b) the oracle in the reference's own order
    (mf_sequential.cu arithmetic, sequential dot product): final test RMSE within 1e-4, parameters within 1e-3."""
    import bench
    tr, te = bench.load_dataset("ml-1m", 20240917, 0, lambda: None)
    f, iters = 50, 2000
    kw = dict(total_iterations=iters, n_factors=f, check_error=500, learning_rate=0.01)
    cfg = cu.default_config(**kw)
    gP, gQ, losses, gub, gib = cu.train(tr, te, cfg, mode="ordered", verbose=False)
    logs = {}
    for name, order in (("tree", orc.DOT_TREE16), ("seq", orc.DOT_SEQ)):
        ocfg = orc.default_config(**kw)
        P, Q, ub, ib = orc.init_model(tr.rows, tr.cols, f)
        logs[name] = (orc.train(_as_orc(tr), _as_orc(te), ocfg, P, Q, ub, ib, tr.global_bias, dot_order=order,
                                acc=orc.ACC_F64, schedule=orc.SCHED_PATIENCE), (P, Q, ub, ib), ocfg.learning_rate)
    log, (P, Q, ub, ib), lr = logs["tree"]
    np.testing.assert_array_equal(gP, P)
    np.testing.assert_array_equal(gQ, Q)
    np.testing.assert_array_equal(gub, ub)
    np.testing.assert_array_equal(gib, ib)
    assert [losses[e["iteration"] - 1] for e in log] == [e["test_rmse"] for e in log] and cfg.learning_rate == lr
    log_s, (Ps, Qs, ubs, ibs), _ = logs["seq"]
    assert abs(float(losses[iters - 1]) - log_s[-1]["test_rmse"]) <= 1e-4
    assert np.abs(gP - Ps).max() <= 1e-3 and np.abs(gQ - Qs).max() <= 1e-3


def test_netflix_shape_full_size_bit_exact():
    """BASELINE.json configs[4] shape (480,189 users x 17,770 items, 79 M integer ratings, f=128 -- the largest
    configuration): 3 ordered iterations and the fused loss against the sequential CPU oracle, every one of the
    63.7 M parameters bit for bit; then Hogwild keeps everything finite and lowers the test RMSE."""
    import bench
    tr, te = bench.load_dataset("netflix", 20240917, 0, lambda: None)
    assert tr.rows == 480189 and tr.cols == 17770 and tr.nnz > 70_000_000
    f = 128
    P, Q, ub, ib = orc.init_model(tr.rows, tr.cols, f)
    model = cu.Model(tr.rows, tr.cols, f, tr.global_bias)
    dtr, dte = cu.DeviceCSR(tr), cu.DeviceCSR(te)
    model.sgd(dtr, HYPER, 42, 0, 3, mode="ordered")
    orc.sgd_iterations(_as_orc(tr), P, Q, ub, ib, tr.global_bias, HYPER, 42, 0, 3, dot_order=orc.DOT_TREE16)
    gP, gQ, gub, gib = model.download()
    np.testing.assert_array_equal(gQ, Q)
    np.testing.assert_array_equal(gP, P)
    np.testing.assert_array_equal(gub, ub)
    np.testing.assert_array_equal(gib, ib)
    got = model.loss(dte)
    want = orc.loss(_as_orc(te), P, Q, ub, ib, tr.global_bias, dot_order=orc.DOT_TREE16)
    assert got["rmse"] == want["rmse"] and got["mae"] == want["mae"]
    # 19.7 M double additions in a different order (per group, per block, blocks) than the oracle's single pass
    assert abs(got["sum_sq"] - want["sum_sq"]) <= 1e-11 * want["sum_sq"]
    model.sgd(dtr, HYPER, 42, 3, 200, mode="hogwild")
    after = model.loss(dte)["rmse"]
    assert np.isfinite(after) and after < got["rmse"]


def test_two_logical_shards_on_one_gpu_match_the_oracle():
    """N logical shards emulated on one device.  Two user shards, each an Engine with its own
    replica of the item side, ordered mode (deterministic), item deltas reconciled every 3 iterations with the
    product's pack / apply kernels (the all-reduce itself replaced by a tensor add) -- against the same schedule run
    by the CPU oracle with numpy doing the merge: every parameter bit for bit."""
    import torch
    from cu2rec_amd.engine import DeviceRatings, Engine
    from cu2rec_amd.sharded import plan_users
    tr, _ = _small_set(users=500, items=150, nnz=12000, seed=21)
    f, n_shards, sync, total = 24, 2, 3, 9
    P, Q, ub, ib = orc.init_model(tr.rows, tr.cols, f)
    b = plan_users(tr.rows, n_shards)
    bounds = list(zip(b[:-1], b[1:]))
    shards = [tr.slice_users(u0, u1) for u0, u1 in bounds]
    engines = [Engine(u1 - u0, tr.cols, f, tr.global_bias, P[u0:u1], Q, ub[u0:u1], ib) for u0, u1 in bounds]
    ratings = [DeviceRatings(s, engines[0].device) for s in shards]
    for e in engines:
        e.snapshot_items()
    # oracle side: full-size arrays, each shard sees only its users' rows through a masked CSR
    oP = [P.copy() for _ in bounds]
    oub = [ub.copy() for _ in bounds]
    oQ, oib = [Q.copy() for _ in bounds], [ib.copy() for _ in bounds]
    Qb, ibb = Q.copy(), ib.copy()
    masked = []
    for u0, u1 in bounds:
        ip = tr.indptr.copy()
        ip[:u0 + 1] = tr.indptr[u0]
        ip[u1:] = tr.indptr[u1]
        masked.append(orc.CSR(ip, tr.indices, tr.data, tr.rows, tr.cols))
    for it in range(0, total, sync):
        for k, (u0, _) in enumerate(bounds):
            engines[k].sgd(ratings[k], HYPER, 42, it, sync, mode="ordered", user_offset=u0)
            orc.sgd_iterations(masked[k], oP[k], oQ[k], oub[k], oib[k], tr.global_bias, HYPER, 42, it, sync,
                               dot_order=orc.DOT_TREE16)
        bufs = [e.pack_item_delta().clone() for e in engines]
        total_buf = bufs[0] + bufs[1]  # what the all-reduce would deliver
        for e in engines:
            e.exchange.copy_(total_buf)
            e.apply_item_delta(0.5)
        dQ = (oQ[0] - Qb) + (oQ[1] - Qb)
        dib = (oib[0] - ibb) + (oib[1] - ibb)
        Qb, ibb = Qb + np.float32(0.5) * dQ, ibb + np.float32(0.5) * dib
        for k in range(n_shards):
            oQ[k], oib[k] = Qb.copy(), ibb.copy()
    for k, (u0, u1) in enumerate(bounds):
        gP, gQ, gub, gib = engines[k].download()
        np.testing.assert_array_equal(gP, oP[k][u0:u1])
        np.testing.assert_array_equal(gub, oub[k][u0:u1])
        np.testing.assert_array_equal(gQ, Qb)
        np.testing.assert_array_equal(gib, ibb)
