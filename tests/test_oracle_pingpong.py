"""CPU: the oracle's restatement of the reference GPU kernel's own semantics (sgd.cu:22-75 + training.cu:107-171,
`orc_sgd_pingpong_iterations`) against hand-built cases: who the first writer is, the rotation of the thread -> user
map, the swap, and the fall-back of an unsampled item to its value of two iterations ago."""
import numpy as np

from oracle import oracle as orc

HYPER = (0.05, 0.02, 0.03, 0.04, 0.01)


def _one_rating_each(items_of_users, cols, ratings=None):
    n = len(items_of_users)
    return orc.CSR(np.arange(n + 1, dtype=np.int32), np.array(items_of_users, np.int32),
                   np.array(ratings if ratings is not None else [4.0] * n, np.float32), n, cols, 3.0)


def _model(rows, cols, f, seed=0):
    rng = np.random.RandomState(seed)
    return ((rng.randn(rows, f) * 0.3).astype(np.float32), (rng.randn(cols, f) * 0.3).astype(np.float32),
            (rng.randn(rows) * 0.1).astype(np.float32), (rng.randn(cols) * 0.1).astype(np.float32))


def _single_update(csr, u, P, Q, ub, ib, it):
    """What user u alone would do to (P[u], ub[u], Q[item], ib[item]) from these values."""
    P, Q, ub, ib = P.copy(), Q.copy(), ub.copy(), ib.copy()
    orc.sgd_one(csr, u, P, Q, ub, ib, csr.global_bias, HYPER, 42, it)
    return P, Q, ub, ib


def test_first_writer_is_the_lowest_rotated_thread_index_and_the_buffers_swap():
    f = 6
    csr = _one_rating_each([0, 0, 0], cols=2, ratings=[5.0, 1.0, 3.0])  # every user can only sample item 0; item 1: nobody
    P0, Q0, ub0, ib0 = _model(3, 2, f)
    P, Q, ub, ib = P0.copy(), Q0.copy(), ub0.copy(), ib0.copy()
    Qt, ibt = Q0.copy(), ib0.copy()  # training.cu:37,69-70: targets start as copies
    orc.sgd_pingpong_iterations(csr, P, Q, Qt, ub, ib, ibt, 3.0, HYPER, 42, 0, 1)
    # iteration 0: start_user 0 -> thread 0 = user 0 claims item 0; every user computed from the ORIGINAL item row
    for u in range(3):
        Pu, Qu, ubu, ibu = _single_update(csr, u, P0, Q0, ub0, ib0, 0)
        np.testing.assert_array_equal(P[u], Pu[u])
        assert ub[u] == ubu[u]
        if u == 0:  # after the swap the current item side holds user 0's write ...
            np.testing.assert_array_equal(Q[0], Qu[0])
            assert ib[0] == ibu[0]
    np.testing.assert_array_equal(Qt[0], Q0[0])  # ... and the other buffer still the original row
    np.testing.assert_array_equal(Q[1], Q0[1])   # item 1 untouched in both
    np.testing.assert_array_equal(Qt[1], Q0[1])
    # iteration 1: start_user = 250 % 3 = 1 -> thread 0 = user 1 is the early bird now
    P1, Q1, ub1, ib1 = P.copy(), Q.copy(), ub.copy(), ib.copy()
    orc.sgd_pingpong_iterations(csr, P, Q, Qt, ub, ib, ibt, 3.0, HYPER, 42, 1, 1)
    _, Qu, _, ibu = _single_update(csr, 1, P1, Q1, ub1, ib1, 1)
    np.testing.assert_array_equal(Q[0], Qu[0])
    assert ib[0] == ibu[0]
    np.testing.assert_array_equal(Qt[0], Q1[0])


def test_an_item_nobody_samples_falls_back_two_iterations():
    f = 4
    # user 0 rates item 0; in iteration 0 it writes item 0's row into the other buffer; we then remove its rating
    csr = _one_rating_each([0], cols=1)
    empty = orc.CSR(np.zeros(2, np.int32), np.zeros(0, np.int32), np.zeros(0, np.float32), 1, 1, 3.0)
    P, Q, ub, ib = _model(1, 1, f, seed=3)
    Q0 = Q.copy()
    Qt, ibt = Q.copy(), ib.copy()
    orc.sgd_pingpong_iterations(csr, P, Q, Qt, ub, ib, ibt, 3.0, HYPER, 42, 0, 1)
    updated = Q.copy()
    assert not np.array_equal(updated, Q0)
    orc.sgd_pingpong_iterations(empty, P, Q, Qt, ub, ib, ibt, 3.0, HYPER, 42, 1, 1)
    np.testing.assert_array_equal(Q, Q0)        # nobody sampled it: the swap brings the ORIGINAL row back (training.cu:164)
    orc.sgd_pingpong_iterations(empty, P, Q, Qt, ub, ib, ibt, 3.0, HYPER, 42, 2, 1)
    np.testing.assert_array_equal(Q, updated)   # ... and the next swap the updated one


def test_swap_last_and_chunking():
    from cu2rec_amd import synth
    tr, _ = synth.make_ratings(200, 60, 3000, min_degree=3, seed=2)
    csr = orc.CSR(tr.indptr, tr.indices, tr.data, tr.rows, tr.cols, tr.global_bias)
    a = [x.copy() for x in _model(tr.rows, tr.cols, 8, seed=5)]
    b = [x.copy() for x in a]
    Qt_a, ibt_a, Qt_b, ibt_b = a[1].copy(), a[3].copy(), b[1].copy(), b[3].copy()
    orc.sgd_pingpong_iterations(csr, a[0], a[1], Qt_a, a[2], a[3], ibt_a, tr.global_bias, HYPER, 42, 0, 7)
    orc.sgd_pingpong_iterations(csr, b[0], b[1], Qt_b, b[2], b[3], ibt_b, tr.global_bias, HYPER, 42, 0, 3)
    orc.sgd_pingpong_iterations(csr, b[0], b[1], Qt_b, b[2], b[3], ibt_b, tr.global_bias, HYPER, 42, 3, 4, swap_last=False)
    orc.pingpong_swap(b[1], Qt_b, b[3], ibt_b)
    for x, y in zip(a + [Qt_a, ibt_a], b + [Qt_b, ibt_b]):
        np.testing.assert_array_equal(x, y)
