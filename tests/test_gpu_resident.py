"""Resident Hogwild launches (-m gpu): one persistent launch per cu2rec_sgd_update call, user rows in the
register file, a grid-wide barrier where the reference has its kernel boundary (training.cu:107-113).

The barrier is what these tests are about.  Hogwild is racy inside an iteration, so exactness needs inputs
where the race cannot happen: `_collision_free_set` builds ratings for which, in every iteration of the test,
no two users sample the same item -- while the SAME item is sampled by different users in different
iterations.  Then one-launch-per-iteration Hogwild equals the sequential oracle bit for bit, and so must the
resident launch, but only if every item row written in iteration i is visible in iteration i + 1 to whatever
CU / XCD reads it next.  A stale read changes bits.
"""
import numpy as np
import pytest

import cu2rec_amd as cu
from cu2rec_amd import synth
from oracle import oracle as orc

pytestmark = pytest.mark.gpu

HYPER = (0.01, 0.02, 0.02, 0.02, 0.02)
OFF, AUTO, FORCE = 0, 1, 2


class resident_policy:
    def __init__(self, policy):
        self.policy = policy

    def __enter__(self):
        self.prev = cu.lib().cu2rec_hogwild_resident(self.policy)

    def __exit__(self, *exc):
        cu.lib().cu2rec_hogwild_resident(self.prev)


def _as_orc(m):
    return orc.CSR(m.indptr, m.indices, m.data, m.rows, m.cols, m.global_bias)


def _collision_free_set(users, items, degree, iter0, iters, seed, sampler_seed=42, empty_every=0, user_offset=0):
    """CSR (users x items, `degree` ratings per user, a user in `empty_every` has none) such that for every
    iteration in [iter0, iter0 + iters) the items sampled by the users are pairwise distinct."""
    assert items >= 2 * users
    rng = np.random.RandomState(seed)
    deg = np.full(users, degree, np.int64)
    if empty_every:
        deg[::empty_every] = 0
    indptr = np.zeros(users + 1, np.int32)
    np.cumsum(deg, out=indptr[1:])
    nnz = int(indptr[-1])
    indices = np.full(nnz, -1, np.int32)
    used = np.zeros((iters, items), bool)  # used[t, j]: item j is sampled by somebody in iteration iter0 + t
    shared = 0
    for u in range(users):
        lo, hi = int(indptr[u]), int(indptr[u + 1])
        if lo == hi:
            continue
        picks = np.array([orc.sample(sampler_seed, user_offset + u, iter0 + t, lo, hi) for t in range(iters)])
        mine = set()
        for k in range(lo, hi):
            when = np.nonzero(picks == k)[0]
            while True:
                j = int(rng.randint(items))
                if j not in mine and not used[when, j].any():
                    break
            shared += int(used[:, j].any())
            used[when, j] = True
            mine.add(j)
            indices[k] = j
    assert shared > users // 2  # plenty of items are visited by different users in different iterations
    data = rng.randint(1, 6, nnz).astype(np.float32)
    return cu.HostCSR(indptr, indices, data, users, items, 3.0)


def _run_and_compare(m, f, iter0, iters, policy, update_items=True, chunks=None):
    P, Q, ub, ib = orc.init_model(m.rows, m.cols, f)
    model = cu.Model(m.rows, m.cols, f, m.global_bias)
    d = cu.DeviceCSR(m)
    with resident_policy(policy):
        it = iter0
        for n in (chunks or [iters]):
            model.sgd(d, HYPER, 42, it, n, mode="hogwild", update_items=update_items)
            it += n
        assert it == iter0 + iters
    orc.sgd_iterations(_as_orc(m), P, Q, ub, ib, m.global_bias, HYPER, 42, iter0, iters, dot_order=orc.DOT_TREE16,
                       update_items=update_items)
    for name, g, w in zip(("P", "Q", "user_bias", "item_bias"), model.download(), (P, Q, ub, ib)):
        np.testing.assert_array_equal(g, w, err_msg=name)


@pytest.fixture(scope="module")
def ml20m_like():
    # 60,000 users on the 8,192 groups of a 256-CU grid: 8 resident users per group (f = 100: two float4 per lane)
    return _collision_free_set(60000, 120000, 6, 0, 12, seed=1, empty_every=97)


def test_collision_free_set_is_exact_with_one_launch_per_iteration(ml20m_like):
    """Control: on this input the streaming kernel (one launch per iteration) IS the sequential result."""
    _run_and_compare(ml20m_like, 100, 0, 12, OFF)


def test_resident_launch_bit_exact_across_the_grid_barrier(ml20m_like):
    """Twelve iterations in ONE launch: equal to the oracle bit for bit, i.e. no stale item row anywhere."""
    _run_and_compare(ml20m_like, 100, 0, 12, FORCE)


def test_resident_auto_policy_and_chunked_calls(ml20m_like):
    """Default policy takes the resident path here (8 users per group, calls of >= 4 iterations); calls of 5 + 4 + 3
    iterations (the last one streams) continue each other exactly."""
    assert cu.lib().cu2rec_hogwild_resident(-1) == AUTO
    _run_and_compare(ml20m_like, 100, 0, 12, AUTO, chunks=[5, 4, 3])


@pytest.mark.parametrize("f,users", [(50, 40000), (64, 9000), (128, 147000), (160, 30000), (256, 20000), (8, 700),
                                     (100, 200000), (50, 300000), (160, 80000), (256, 64000)])
def test_resident_every_row_width(f, users):
    """J = 1..4 float4 slots per lane, few and many users per group (147,000 users at f = 128 is the 18 per group the
    registers hold; 200,000 users at f = 100 need 25 per group, 9 of them in LDS; likewise the last three cases), a
    grid smaller than the chip (700 users), resume from a non-zero iteration."""
    m = _collision_free_set(users, 2 * users + 64, 4, 7, 6, seed=f, empty_every=0 if users < 1000 else 41)
    with resident_policy(FORCE):
        assert cu.lib().cu2rec_hogwild_resident_plan(users, f, 6, None, None) == 1
    _run_and_compare(m, f, 7, 6, FORCE)


@pytest.mark.parametrize("f,users,streamed", [(128, 230000, 9), (100, 300000, 17), (64, 450000, 16), (256, 100000, 5), (160, 140000, 7)])
def test_partial_residency_bit_exact_across_the_grid_barrier(f, users, streamed):
    """Round 4: a set that does not fit the chip still runs as ONE launch per call -- every group's first rows resident, `streamed`
    more per group passing through the same pipeline from and to memory in chunks of 16 (one chunk, a chunk and a bit, exactly one
    chunk; J = 1..4).  Bit for bit the oracle on a collision-free set, resumed from a non-zero iteration, users without ratings
    among the streamed ones."""
    import ctypes as C
    assert cu.lib().cu2rec_hogwild_resident_streamed_rows(users, f, 256) == streamed
    m = _collision_free_set(users, 2 * users + 64, 3, 5, 5, seed=f + 1, empty_every=53)
    with resident_policy(FORCE):
        blocks, rows = C.c_int(0), C.c_int(0)
        assert cu.lib().cu2rec_hogwild_resident_plan(users, f, 5, C.byref(blocks), C.byref(rows)) == 1
        assert rows.value * blocks.value * 32 >= users
    _run_and_compare(m, f, 5, 5, FORCE, chunks=[3, 2])


def test_resident_frozen_items_need_no_barrier():
    """is_train == false: item side untouched, users fit as in predict.cu; equal to the oracle on ANY input
    because nothing is shared between users."""
    tr, _ = synth.make_ratings(30000, 500, 400000, min_degree=3, seed=5)
    _run_and_compare(tr, 100, 3, 9, FORCE, update_items=False)


def test_resident_falls_back_when_the_rows_do_not_fit():
    """More users than the register file holds (f = 256: 9 per group): the call streams instead, same results."""
    m = _collision_free_set(80000, 160064, 3, 0, 4, seed=9)
    _run_and_compare(m, 256, 0, 4, FORCE)


def test_resident_one_iteration_is_jacobi():
    """The Hogwild contract inside an iteration (as for the streaming kernel): users / items touched by exactly one
    update equal the oracle's single update bit for bit; untouched items keep their bits."""
    tr, _ = synth.make_ratings(40000, 3000, 600000, min_degree=3, seed=9)
    f = 100
    rng = np.random.RandomState(2)
    P0, Q0 = (rng.randn(tr.rows, f) * 0.1).astype(np.float32), (rng.randn(tr.cols, f) * 0.1).astype(np.float32)
    ub0, ib0 = (rng.randn(tr.rows) * 0.1).astype(np.float32), (rng.randn(tr.cols) * 0.1).astype(np.float32)
    model = cu.Model(tr.rows, tr.cols, f, tr.global_bias, P0, Q0, ub0, ib0)
    with resident_policy(FORCE):
        model.sgd(cu.DeviceCSR(tr), HYPER, 42, 5, 1, mode="hogwild")
    gP, gQ, gub, gib = model.download()
    items = np.array([tr.indices[orc.sample(42, u, 5, tr.indptr[u], tr.indptr[u + 1])] for u in range(tr.rows)])
    counts = np.bincount(items, minlength=tr.cols)
    o = _as_orc(tr)
    singles = np.nonzero(counts[items] == 1)[0]
    assert len(singles) > 20
    for u in singles[:400]:
        P, Q, ub, ib = P0.copy(), Q0.copy(), ub0.copy(), ib0.copy()
        orc.sgd_one(o, int(u), P, Q, ub, ib, tr.global_bias, HYPER, 42, 5, dot_order=orc.DOT_TREE16)
        np.testing.assert_array_equal(gP[u], P[u])
        np.testing.assert_array_equal(gQ[items[u]], Q[items[u]])
        assert gub[u] == ub[u] and gib[items[u]] == ib[items[u]]
    np.testing.assert_array_equal(gQ[counts == 0], Q0[counts == 0])
    assert np.isfinite(gP).all() and np.isfinite(gQ).all()


def test_resident_converges_like_streaming_and_the_oracle():
    """Real (colliding) ratings: resident and streaming Hogwild are the same algorithm, so 300 iterations end at the
    same test RMSE up to the outcome of the race -- which users of an iteration overlap in time differs (a resident
    grid has ALL of them in flight at once, a streamed iteration runs in waves of workgroups), so popular items lose
    a few more or a few fewer updates: 5e-3 here, 1e-4 on the ML-20M shape.  Both within 2e-2 of the
    sequential oracle."""
    tr, te = synth.make_ratings(40000, 6000, 1200000, min_degree=5, seed=11)
    f, iters = 32, 300
    P, Q, ub, ib = orc.init_model(tr.rows, tr.cols, f)
    start = orc.loss(_as_orc(te), P, Q, ub, ib, tr.global_bias)["rmse"]
    orc.sgd_iterations(_as_orc(tr), P, Q, ub, ib, tr.global_bias, HYPER, 42, 0, iters, dot_order=orc.DOT_TREE16)
    want = orc.loss(_as_orc(te), P, Q, ub, ib, tr.global_bias)["rmse"]
    d_tr, d_te = cu.DeviceCSR(tr), cu.DeviceCSR(te)
    got = {}
    for policy in (OFF, FORCE):
        model = cu.Model(tr.rows, tr.cols, f, tr.global_bias)
        with resident_policy(policy):
            model.sgd(d_tr, HYPER, 42, 0, iters, mode="hogwild")
        got[policy] = model.loss(d_te)["rmse"]
    assert want < start - 0.03
    assert abs(got[FORCE] - got[OFF]) < 5e-3
    assert abs(got[FORCE] - want) < 2e-2 and abs(got[OFF] - want) < 2e-2


@pytest.mark.parametrize("with_pairs", [True, False])
def test_resident_through_the_raw_pointer_abi_with_user_offset(with_pairs):
    """cu2rec_sgd_update_ex on torch-owned buffers (the multi-GPU plumbing): a shard whose users are global ids
    [u0, u0 + n) draws what the unsharded run draws for them -- with the side-by-side sample array
    (cu2rec_sample_pairs_build, one 8-byte gather per draw) and without it (two 4-byte gathers): same bits."""
    from cu2rec_amd import api
    from cu2rec_amd.engine import DeviceRatings, Engine
    u0, n, f = 5000, 50000, 100
    m = _collision_free_set(n, 2 * n + 64, 4, 2, 8, seed=3, user_offset=u0)
    P, Q, ub, ib = orc.init_model(n, m.cols, f)
    eng = Engine(n, m.cols, f, m.global_bias, P.copy(), Q.copy(), ub.copy(), ib.copy())
    d = DeviceRatings(m, eng.device)
    with resident_policy(FORCE):
        if with_pairs:
            eng.sgd(d, HYPER, 42, 2, 8, cu.SGD_HOGWILD, True, u0)
            assert d._pairs is not None
            packed = d._pairs.cpu().numpy()
            np.testing.assert_array_equal((packed & 0xFFFFFFFF).astype(np.int32), m.indices)
            np.testing.assert_array_equal((packed >> 32).astype(np.uint32).view(np.float32), m.data)
        else:
            api.sgd_update(d.indptr.data_ptr(), d.indices.data_ptr(), d.data.data_ptr(), n, m.cols, eng.P.data_ptr(), eng.ld,
                           eng.Q.data_ptr(), eng.ldq, eng.user_bias.data_ptr(), eng.item_bias.data_ptr(), eng.global_bias, f,
                           HYPER, 42, 2, 8, cu.SGD_HOGWILD, True, u0, None, None)
    # oracle: the same users placed at their global ids behind u0 empty users
    big = cu.HostCSR(np.concatenate([np.zeros(u0, np.int32), m.indptr]), m.indices, m.data, u0 + n, m.cols, m.global_bias)
    Pb, ubb = np.zeros((u0 + n, f), np.float32), np.zeros(u0 + n, np.float32)
    Pb[u0:], ubb[u0:] = P, ub
    orc.sgd_iterations(_as_orc(big), Pb, Q, ubb, ib, m.global_bias, HYPER, 42, 2, 8, dot_order=orc.DOT_TREE16)
    gP, gQ, gub, gib = eng.download()
    np.testing.assert_array_equal(gP, Pb[u0:])
    np.testing.assert_array_equal(gQ, Q)
    np.testing.assert_array_equal(gub, ubb[u0:])
    np.testing.assert_array_equal(gib, ib)


def test_train_loop_with_resident_segments_equals_the_oracle(ml20m_like):
    """cu2rec_train (training.cu:21-204) in Hogwild mode: the stretches between two loss checks (1, 3, 4, 4
    iterations for check_error = 4) run as streaming launches and as resident launches (default policy: >= 4
    iterations per call); on the collision-free input the whole run -- factors, the loss at every check, the learning
    rate after patience decay -- equals the oracle's train() bit for bit."""
    m = ml20m_like  # collision free for iterations 0..11
    rng = np.random.RandomState(7)
    te = cu.HostCSR(np.arange(0, 3 * 2000 + 1, 3, dtype=np.int32), rng.randint(0, m.cols, 6000).astype(np.int32),
                    rng.randint(1, 6, 6000).astype(np.float32), 2000, m.cols, m.global_bias)
    kw = dict(total_iterations=12, n_factors=100, check_error=4, learning_rate=0.05, patience=1.0, seed=42)
    cfg, ocfg = cu.default_config(**kw), orc.default_config(**kw)
    P, Q, ub, ib = orc.init_model(m.rows, m.cols, 100)
    log = orc.train(_as_orc(m), _as_orc(te), ocfg, P, Q, ub, ib, m.global_bias, dot_order=orc.DOT_TREE16,
                    acc=orc.ACC_F64, schedule=orc.SCHED_PATIENCE)
    assert cu.lib().cu2rec_hogwild_resident(-1) == AUTO
    gP, gQ, losses, gub, gib = cu.train(m, te, cfg, mode="hogwild", verbose=False)
    assert [e["iteration"] for e in log] == [1, 4, 8, 12]  # training.cu:118
    np.testing.assert_array_equal(gP, P)
    np.testing.assert_array_equal(gQ, Q)
    np.testing.assert_array_equal(gub, ub)
    np.testing.assert_array_equal(gib, ib)
    for e in log:
        assert losses[e["iteration"] - 1] == e["test_rmse"]
    assert cfg.learning_rate == ocfg.learning_rate and cfg.cur_iterations == 12


@pytest.mark.parametrize("cols", [1, 0])
def test_resident_nobody_has_ratings(cols):
    """Edge: 40,000 users, no rating at all (sgd.cu:34 skips every one of them).  With one item the forced resident
    launch runs (every user is a sink user); with no item at all the call streams.  Either way nothing changes."""
    rows, f = 40000, 100
    m = cu.HostCSR(np.zeros(rows + 1, np.int32), np.zeros(0, np.int32), np.zeros(0, np.float32), rows, cols, 3.0)
    model = cu.Model(rows, cols, f, 3.0)
    before = model.download()
    with resident_policy(FORCE):
        model.sgd(cu.DeviceCSR(m), HYPER, 42, 0, 6, mode="hogwild")
    for b, a in zip(before, model.download()):
        np.testing.assert_array_equal(a, b)
