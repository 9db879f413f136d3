"""CU2REC_SGD_PINGPONG (-m gpu): the reference GPU kernel's own semantics -- item side read as of the start of the
iteration, first claimant writes the second buffer pair, swap after every iteration (sgd.cu:22-75,
training.cu:107-171) -- against the oracle's restatement (tests/test_oracle_pingpong.py pins that one on the CPU).
The mode is deterministic (first writer = lowest rotated thread index, decided by a 64-bit atomicMin), so every
comparison is bit for bit."""
import ctypes as C
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


def _as_orc(m):
    return orc.CSR(m.indptr, m.indices, m.data, m.rows, m.cols, m.global_bias)


def _set(users=3000, items=400, nnz=60000, seed=4, empty_every=37):
    tr, te = synth.make_ratings(users, items, nnz, min_degree=3, seed=seed)
    if empty_every:  # some users without ratings (sgd.cu:34)
        deg = np.diff(tr.indptr)
        keep = np.ones(tr.nnz, bool)
        for u in range(0, users, empty_every):
            keep[tr.indptr[u]:tr.indptr[u + 1]] = False
            deg[u] = 0
        indptr = np.zeros(users + 1, np.int32)
        np.cumsum(deg, out=indptr[1:])
        tr = cu.HostCSR(indptr, tr.indices[keep], tr.data[keep], users, items, tr.global_bias)
    return tr, te


@pytest.mark.parametrize("f,iters,iter0", [(10, 1, 0), (100, 4, 0), (100, 7, 13), (300, 3, 2), (64, 6, 0)])
def test_pingpong_bit_exact_vs_oracle(f, iters, iter0):
    tr, _ = _set()
    P, Q, ub, ib = orc.init_model(tr.rows, tr.cols, f)
    Qt, ibt = Q.copy(), ib.copy()
    model = cu.Model(tr.rows, tr.cols, f, tr.global_bias)
    model.sgd(cu.DeviceCSR(tr), HYPER, 42, iter0, iters, mode="pingpong")
    orc.sgd_pingpong_iterations(_as_orc(tr), P, Q, Qt, ub, ib, ibt, tr.global_bias, HYPER, 42, iter0, iters,
                                dot_order=orc.DOT_TREE16)
    for name, g, w in zip(("P", "Q", "user_bias", "item_bias"), model.download(), (P, Q, ub, ib)):
        np.testing.assert_array_equal(g, w, err_msg=name)


def test_pingpong_calls_continue_each_other_and_are_reproducible():
    tr, _ = _set(seed=6)
    f = 50
    d = cu.DeviceCSR(tr)
    a, b, c = (cu.Model(tr.rows, tr.cols, f, tr.global_bias) for _ in range(3))
    a.sgd(d, HYPER, 42, 0, 9, mode="pingpong")
    b.sgd(d, HYPER, 42, 0, 4, mode="pingpong")
    b.sgd(d, HYPER, 42, 4, 5, mode="pingpong")  # the second buffer pair carries over
    c.sgd(d, HYPER, 42, 0, 9, mode="pingpong")
    for x, y, z in zip(a.download(), b.download(), c.download()):
        np.testing.assert_array_equal(x, y)
        np.testing.assert_array_equal(x, z)


def test_pingpong_frozen_items():
    tr, _ = _set(seed=8)
    f = 20
    P, Q, ub, ib = orc.init_model(tr.rows, tr.cols, f)
    Qt, ibt = Q.copy(), ib.copy()
    Q0 = Q.copy()
    model = cu.Model(tr.rows, tr.cols, f, tr.global_bias)
    model.sgd(cu.DeviceCSR(tr), HYPER, 42, 0, 5, mode="pingpong", update_items=False)
    orc.sgd_pingpong_iterations(_as_orc(tr), P, Q, Qt, ub, ib, ibt, tr.global_bias, HYPER, 42, 0, 5,
                                dot_order=orc.DOT_TREE16, update_items=False)
    gP, gQ, gub, gib = model.download()
    np.testing.assert_array_equal(gQ, Q0)
    np.testing.assert_array_equal(gP, P)
    np.testing.assert_array_equal(gub, ub)


def test_pingpong_train_evaluates_the_loss_before_the_swap():
    """train() in this mode: the loss of a check is taken on this iteration's P and the item side the iteration READ
    (training.cu:121 runs before the swap at :164); emulated here with the oracle's pieces."""
    tr, te = _set(seed=9, empty_every=0)
    f = 16
    kw = dict(total_iterations=11, n_factors=f, check_error=4, learning_rate=0.05, patience=1.0, seed=42)
    cfg = cu.default_config(**kw)
    gP, gQ, losses, gub, gib = cu.train(tr, te, cfg, mode="pingpong", verbose=False)
    P, Q, ub, ib = orc.init_model(tr.rows, tr.cols, f)
    Qt, ibt = Q.copy(), ib.copy()
    lr, patience, last = np.float32(0.05), 1, np.float32(np.finfo(np.float32).max)
    i, want = 0, {}
    for seg_end in (0, 3, 7, 10):  # training.cu:118 with check_error 4, 11 iterations
        n = seg_end - i + 1
        hyper = (float(lr), 0.02, 0.02, 0.02, 0.02)
        orc.sgd_pingpong_iterations(_as_orc(tr), P, Q, Qt, ub, ib, ibt, tr.global_bias, hyper, 42, i, n,
                                    dot_order=orc.DOT_TREE16, swap_last=False)
        rmse = np.float32(orc.loss(_as_orc(te), P, Q, ub, ib, tr.global_bias, dot_order=orc.DOT_TREE16)["rmse"])
        orc.pingpong_swap(Q, Qt, ib, ibt)
        want[seg_end] = rmse
        if last < rmse:
            patience -= 1
        if patience <= 0:
            patience, lr = 1, np.float32(lr * np.float32(0.2))
        last, i = rmse, seg_end + 1
    for k, v in want.items():
        assert losses[k] == v, (k, losses[k], v)
    np.testing.assert_array_equal(gP, P)
    np.testing.assert_array_equal(gQ, Q)
    np.testing.assert_array_equal(gib, ib)
    assert np.float32(cfg.learning_rate) == lr


def test_pingpong_raw_pointer_abi_reports_the_swap():
    """cu2rec_sgd_update_pingpong on caller-owned buffers (sgd.h:12-16 with Q_target / item_bias_target): after an odd
    number of iterations the current item side is in the target buffers and *swapped says so."""
    import torch
    from cu2rec_amd._lib import Hyper, check, lib
    from cu2rec_amd.engine import DeviceRatings, Engine
    tr, _ = _set(users=1500, items=300, nnz=30000, seed=10)
    f = 100
    P, Q, ub, ib = orc.init_model(tr.rows, tr.cols, f)
    eng = Engine(tr.rows, tr.cols, f, tr.global_bias, P.copy(), Q.copy(), ub.copy(), ib.copy())
    d = DeviceRatings(tr, eng.device)
    Q_t, ib_t = eng.Q.clone(), eng.item_bias.clone()
    claim = torch.empty(tr.cols, dtype=torch.int64, device=eng.device)
    swapped = C.c_int(-1)
    h = Hyper(*HYPER)
    check(lib().cu2rec_sgd_update_pingpong(d.indptr.data_ptr(), d.indices.data_ptr(), d.data.data_ptr(), tr.rows, tr.cols,
                                           eng.P.data_ptr(), eng.ld, eng.Q.data_ptr(), Q_t.data_ptr(), eng.ldq,
                                           eng.user_bias.data_ptr(), eng.item_bias.data_ptr(), ib_t.data_ptr(),
                                           claim.data_ptr(), eng.global_bias, f, C.byref(h), 42, 0, 5, 1, 0, 1,
                                           C.byref(swapped), None))
    assert swapped.value == 1
    Qt, ibt = Q.copy(), ib.copy()
    orc.sgd_pingpong_iterations(_as_orc(tr), P, Q, Qt, ub, ib, ibt, tr.global_bias, HYPER, 42, 0, 5, dot_order=orc.DOT_TREE16)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(Q_t[:tr.cols, :f].cpu().numpy(), Q)       # current side: in the "target" buffers
    np.testing.assert_array_equal(eng.Q[:tr.cols, :f].cpu().numpy(), Qt)    # the other side: in the "Q" buffers
    np.testing.assert_array_equal(ib_t[:tr.cols].cpu().numpy(), ib)
    np.testing.assert_array_equal(eng.P[:tr.rows, :f].cpu().numpy(), P)


def test_bin_mf_pingpong(tmp_path):
    exe = os.path.join(ROOT, "bin", "mf")
    train = tmp_path / "ratings.csv"
    train.write_text(open(os.path.join(GOLDEN, "toy_ratings.csv")).read())
    cfgp = tmp_path / "c.cfg"
    cfgp.write_text("0 20 4 0.01 42 0.02 0.02 0.02 0.02\n")
    out = subprocess.run([exe, "-c", str(cfgp), "-m", "pingpong", str(train), os.path.join(GOLDEN, "toy_ratings2.csv")],
                         stdout=subprocess.PIPE, text=True, check=True).stdout
    assert "TEST: Iteration 20 GPU MAE:" in out
    tr = cu.createSparseMatrix(str(train))
    te = cu.createSparseMatrix(os.path.join(GOLDEN, "toy_ratings2.csv"))
    cfg = cu.default_config(total_iterations=20, n_factors=4, learning_rate=0.01, seed=42)
    P, Q, _, ub, ib = cu.train(tr, te, cfg, mode="pingpong", verbose=False)
    got = np.loadtxt(str(tmp_path / "ratings_f4_q.csv"), delimiter=",", dtype=np.float64)
    np.testing.assert_allclose(got, Q, atol=5e-7)  # "%f": six decimals
