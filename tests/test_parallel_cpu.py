"""The N>1 host logic on CPU: user sharding + the item-factor exchange over torch.distributed (gloo,
world_size 2), with an oracle-backed stand-in engine (tests may use the oracle; the product engine is
HIP-only).  Checks: cadence bookkeeping, same sampler stream as the unsharded run, the merge algebra
Q = Q_base + scale * sum_k (Q_k - Q_base), global loss reduction."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import cu2rec_amd as cu
from cu2rec_amd import synth
from exchange_reference import ShardedSGD, plan_users
from conftest import ROOT
from oracle import oracle as orc

HYPER = (0.01, 0.02, 0.02, 0.02, 0.02)


class OracleEngine:
    """Same duck-typed surface as cu2rec_amd.engine.Engine, computed by the CPU oracle (test only)."""

    def __init__(self, rows, cols, f, global_bias, P, Q, ub, ib):
        self.P, self.Q, self.ub, self.ib = P.copy(), Q.copy(), ub.copy(), ib.copy()
        self.f, self.global_bias = f, global_bias
        self.device = None

    def sgd(self, ratings, hyper, seed, iter0, n_iters, mode, update_items=True, user_offset=0):
        # the oracle keys its sampler by row index; emulate a global user id by shifting rows
        o = orc.CSR(ratings.indptr, ratings.indices, ratings.data, ratings.rows, ratings.cols)
        for i in range(n_iters):
            for x in range(ratings.rows):
                lo, hi = int(o.indptr[x]), int(o.indptr[x + 1])
                if lo == hi:
                    continue
                # one update with the draw of (user_offset + x): temporarily present a 1-row view
                y_i = orc.sample(seed, user_offset + x, iter0 + i, lo, hi)
                self._update(x, int(o.indices[y_i]), float(o.data[y_i]), hyper, update_items)

    def _update(self, x, y, r, h, update_items):
        lr, pr, qr, ur, ir = (np.float32(v) for v in h)
        p, q = self.P[x].copy(), self.Q[y].copy()
        pred = orc.lib().orc_predict(self.f, p.ctypes.data_as(orc.C.POINTER(orc.C.c_float)),
                                     q.ctypes.data_as(orc.C.POINTER(orc.C.c_float)), float(self.ub[x]),
                                     float(self.ib[y]), float(self.global_bias), orc.DOT_SEQ)
        err = np.float32(r) - np.float32(pred)
        self.P[x] = p + lr * (err * q - pr * p)
        if update_items:
            self.Q[y] = q + lr * (err * p - qr * q)
            self.ib[y] = self.ib[y] + lr * (err - ir * self.ib[y])
        self.ub[x] = self.ub[x] + lr * (err - ur * self.ub[x])

    def snapshot_items(self):
        self.Q_base, self.ib_base = self.Q.copy(), self.ib.copy()

    def pack_item_delta(self, item_weight=None):
        dQ, dib = self.Q - self.Q_base, self.ib - self.ib_base
        if item_weight is not None:
            w = item_weight.numpy().astype(np.float32)
            dQ, dib = dQ * w[:, None], dib * w
        self._buf = torch.from_numpy(np.concatenate([dQ.ravel(), dib]).astype(np.float32))
        return self._buf

    def snapshot_for_overlap(self):
        self.Q_snap, self.ib_snap = self.Q.copy(), self.ib.copy()
        self.exchange = self._buf

    def apply_item_delta_overlapped(self, scale):
        d = self.exchange.numpy() * np.float32(scale)
        n = self.Q.size
        mQ, mib = self.Q_base + d[:n].reshape(self.Q.shape), self.ib_base + d[n:]
        self.Q, self.ib = mQ + (self.Q - self.Q_snap), mib + (self.ib - self.ib_snap)
        self.Q_base, self.ib_base = mQ, mib

    def apply_item_delta(self, scale):
        d = self._buf.numpy() * np.float32(scale)
        n = self.Q.size
        self.Q = self.Q_base + d[:n].reshape(self.Q.shape)
        self.ib = self.ib_base + d[n:]
        self.snapshot_items()

    def loss(self, ratings):
        o = orc.CSR(ratings.indptr, ratings.indices, ratings.data, ratings.rows, ratings.cols)
        return orc.loss(o, self.P, self.Q, self.ub, self.ib, self.global_bias)

    def download(self):
        return self.P.copy(), self.Q.copy(), self.ub.copy(), self.ib.copy()


def test_single_rank_cadence_terminates_and_matches_oracle():
    tr, te = synth.make_ratings(60, 30, 600, min_degree=2, seed=2)
    P, Q, ub, ib = orc.init_model(tr.rows, tr.cols, 8)
    eng = OracleEngine(tr.rows, tr.cols, 8, tr.global_bias, P, Q, ub, ib)
    job = ShardedSGD(eng, tr, sync_every=3)
    assert job.run(HYPER, 42, 0, 10, cu.SGD_HOGWILD) == 10  # crosses several sync points with world_size 1
    orc.sgd_iterations(orc.CSR(tr.indptr, tr.indices, tr.data, tr.rows, tr.cols), P, Q, ub, ib, tr.global_bias, HYPER,
                       42, 0, 10)
    np.testing.assert_allclose(eng.P, P, atol=1e-6)
    np.testing.assert_allclose(eng.Q, Q, atol=1e-6)
    assert job.exchanges == 0
    assert job.loss(te)["n"] == te.nnz


def test_frozen_items_run_longer_than_a_period_terminates():
    """with update_items == False nothing is exchanged, but the period counter must still start over --
    otherwise the loop asks the engine for 0 iterations forever once n_iters exceeds sync_every."""
    tr, _ = synth.make_ratings(60, 30, 600, min_degree=2, seed=2)
    P, Q, ub, ib = orc.init_model(tr.rows, tr.cols, 8)

    class Counting(OracleEngine):
        calls = []

        def sgd(self, ratings, hyper, seed, iter0, n_iters, mode, update_items=True, user_offset=0):
            assert n_iters > 0 and len(self.calls) < 50, "the loop must make progress"
            self.calls.append((iter0, n_iters))
            super().sgd(ratings, hyper, seed, iter0, n_iters, mode, update_items, user_offset)

    eng = Counting(tr.rows, tr.cols, 8, tr.global_bias, P, Q, ub, ib)
    job = ShardedSGD(eng, tr, sync_every=10)
    assert job.run(HYPER, 42, 0, 25, cu.SGD_HOGWILD, update_items=False) == 25
    assert eng.calls == [(0, 10), (10, 10), (20, 5)] and job.exchanges == 0
    np.testing.assert_array_equal(eng.Q, Q)  # frozen
    orc.sgd_iterations(orc.CSR(tr.indptr, tr.indices, tr.data, tr.rows, tr.cols), P, Q, ub, ib, tr.global_bias, HYPER,
                       42, 0, 25, update_items=False)
    np.testing.assert_allclose(eng.P, P, atol=1e-6)


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tr, te = synth.make_ratings(80, 25, 900, min_degree=2, seed=5)
    f = 6
    P, Q, ub, ib = orc.init_model(tr.rows, tr.cols, f)
    bounds = plan_users(tr.rows, world)
    u0, u1 = bounds[rank], bounds[rank + 1]
    shard, shard_te = tr.slice_users(u0, u1), te.slice_users(u0, u1)
    eng = OracleEngine(u1 - u0, tr.cols, f, tr.global_bias, P[u0:u1], Q, ub[u0:u1], ib)
    job = ShardedSGD(eng, shard, user_offset=u0, sync_every=4, merge="mean")
    job.run(HYPER, 42, 0, 10, cu.SGD_HOGWILD)  # exchanges after iterations 4 and 8
    res = job.loss(shard_te)
    np.savez(os.path.join(out_dir, "r%d.npz" % rank), P=eng.P, Q=eng.Q, ub=eng.ub, ib=eng.ib, u0=u0, u1=u1,
             exchanges=job.exchanges, rmse=res["rmse"], n=res["n"])
    dist.destroy_process_group()


def test_two_ranks_gloo(tmp_path):
    world, port = 2, 29000 + os.getpid() % 2000
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r = [np.load(str(tmp_path / ("r%d.npz" % k))) for k in range(world)]
    assert int(r[0]["exchanges"]) == 2 and int(r[1]["exchanges"]) == 2
    # single-process emulation of the same schedule: two logical shards, merge by mean of deltas
    tr, te = synth.make_ratings(80, 25, 900, min_degree=2, seed=5)
    f = 6
    P, Q, ub, ib = orc.init_model(tr.rows, tr.cols, f)
    bounds = plan_users(tr.rows, world)
    engs = [OracleEngine(bounds[k + 1] - bounds[k], tr.cols, f, tr.global_bias, P[bounds[k]:bounds[k + 1]], Q,
                         ub[bounds[k]:bounds[k + 1]], ib) for k in range(world)]
    shards = [tr.slice_users(bounds[k], bounds[k + 1]) for k in range(world)]
    Qb, ibb = Q.copy(), ib.copy()
    it = 0
    for n in (4, 4, 2):
        for k in range(world):
            engs[k].sgd(shards[k], HYPER, 42, it, n, cu.SGD_HOGWILD, True, bounds[k])
        it += n
        if n == 4:
            dQ = sum(e.Q - Qb for e in engs) / np.float32(world)
            dib = sum(e.ib - ibb for e in engs) / np.float32(world)
            Qb, ibb = Qb + dQ, ibb + dib
            for e in engs:
                e.Q, e.ib = Qb.copy(), ibb.copy()
    for k in range(world):
        np.testing.assert_allclose(r[k]["P"], engs[k].P, atol=1e-6)
        np.testing.assert_allclose(r[k]["Q"], engs[k].Q, atol=1e-6)
        np.testing.assert_allclose(r[k]["ib"], engs[k].ib, atol=1e-6)
    # after the last exchange both replicas agreed; they then drifted for 2 local iterations
    assert float(r[0]["rmse"]) == float(r[1]["rmse"]) and int(r[0]["n"]) == te.nnz  # global loss is all-reduced


def _oracle_factory(rows, cols, f, gb, P0, ub0, tr, te):
    Q = orc.normal_fill(cols * f, f).reshape(cols, f)
    ib = orc.normal_fill(cols, f)
    return OracleEngine(rows, cols, f, gb, P0, Q, ub0, ib), tr, te


def test_train_sharded_single_rank_matches_oracle_train(capsys):
    """train_sharded with one rank: the training.cu schedule (cadence, patience / LR decay, printed lines)."""
    from exchange_reference import train_sharded
    tr, te = synth.make_ratings(60, 30, 700, min_degree=2, seed=4)
    cfg = cu.default_config(total_iterations=20, n_factors=6, check_error=5, learning_rate=0.05, patience=1.0)
    ocfg = orc.default_config(total_iterations=20, n_factors=6, check_error=5, learning_rate=0.05, patience=1.0)
    P, Q, ub, ib = orc.init_model(tr.rows, tr.cols, 6)
    log = orc.train(orc.CSR(tr.indptr, tr.indices, tr.data, tr.rows, tr.cols), orc.CSR(te.indptr, te.indices, te.data, te.rows, te.cols),
                    ocfg, P, Q, ub, ib, tr.global_bias, dot_order=orc.DOT_SEQ, acc=orc.ACC_F64, schedule=orc.SCHED_PATIENCE)
    gP, gQ, losses, gub, gib, (u0, u1) = train_sharded(tr, te, cfg, engine_factory=_oracle_factory, verbose=True)
    assert (u0, u1) == (0, tr.rows)
    np.testing.assert_allclose(gP, P, atol=1e-6)
    np.testing.assert_allclose(gQ, Q, atol=1e-6)
    assert [e["iteration"] - 1 for e in log] == [i for i in range(20) if not np.isnan(losses[i])] == [0, 4, 9, 14, 19]
    for e in log:
        assert abs(losses[e["iteration"] - 1] - e["test_rmse"]) < 1e-6
    assert abs(cfg.learning_rate - ocfg.learning_rate) < 1e-9 and cfg.cur_iterations == 20
    out = capsys.readouterr().out
    assert "TRAIN: Iteration 1 GPU MAE:" in out and "TEST: Iteration 20 GPU MAE:" in out and "Time taken for 20 of iterations" in out


def _train_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from exchange_reference import train_sharded
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tr, te = synth.make_ratings(90, 25, 1000, min_degree=2, seed=6)
    cfg = cu.default_config(total_iterations=12, n_factors=5, check_error=4, learning_rate=0.02)
    P, Q, losses, ub, ib, (u0, u1) = train_sharded(tr, te, cfg, sync_every=3, engine_factory=_oracle_factory, verbose=False)
    np.savez(os.path.join(out_dir, "t%d.npz" % rank), Q=Q, ib=ib, losses=losses, u0=u0, u1=u1, lr=cfg.learning_rate)
    dist.destroy_process_group()


def test_train_sharded_two_ranks_gloo(tmp_path):
    world, port = 2, 31000 + os.getpid() % 2000
    mp.spawn(_train_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    a, b = (np.load(str(tmp_path / ("t%d.npz" % k))) for k in range(2))
    assert (int(a["u0"]), int(a["u1"]), int(b["u0"]), int(b["u1"])) == (0, 45, 45, 90)
    # after the final exchange both ranks hold the same item side and saw the same global losses / LR
    np.testing.assert_array_equal(a["Q"], b["Q"])
    np.testing.assert_array_equal(a["ib"], b["ib"])
    np.testing.assert_array_equal(np.nan_to_num(a["losses"]), np.nan_to_num(b["losses"]))
    assert float(a["lr"]) == float(b["lr"])
    checked = [i for i in range(12) if not np.isnan(a["losses"][i])]
    assert checked == [0, 3, 7, 11] and a["losses"][11] < a["losses"][0]


def _weighted_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tr, _ = synth.make_ratings(80, 25, 900, min_degree=2, seed=5)
    f = 6
    P, Q, ub, ib = orc.init_model(tr.rows, tr.cols, f)
    bounds = plan_users(tr.rows, world)
    u0, u1 = bounds[rank], bounds[rank + 1]
    shard = tr.slice_users(u0, u1)
    eng = OracleEngine(u1 - u0, tr.cols, f, tr.global_bias, P[u0:u1], Q, ub[u0:u1], ib)
    job = ShardedSGD(eng, shard, user_offset=u0, sync_every=5, merge="weighted", item_rates=cu.api.item_update_rates(shard))
    job.run(HYPER, 42, 0, 5, cu.SGD_HOGWILD)
    np.savez(os.path.join(out_dir, "w%d.npz" % rank), Q=eng.Q, ib=eng.ib, w=job.item_weight.numpy())
    dist.destroy_process_group()


def test_weighted_merge_two_ranks_gloo(tmp_path):
    """merge='weighted': per-item weights rate_k / sum rate (they sum to one over ranks), result = Q_base + sum_k w_k delta_k."""
    world, port = 2, 33000 + os.getpid() % 2000
    mp.spawn(_weighted_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    a, b = (np.load(str(tmp_path / ("w%d.npz" % k))) for k in range(2))
    np.testing.assert_allclose(a["w"] + b["w"], 1.0, atol=1e-6)
    np.testing.assert_array_equal(a["Q"], b["Q"])
    tr, _ = synth.make_ratings(80, 25, 900, min_degree=2, seed=5)
    f = 6
    P, Q, ub, ib = orc.init_model(tr.rows, tr.cols, f)
    bounds = plan_users(tr.rows, world)
    deltas, rates = [], []
    for k in range(world):
        sh = tr.slice_users(bounds[k], bounds[k + 1])
        e = OracleEngine(bounds[k + 1] - bounds[k], tr.cols, f, tr.global_bias, P[bounds[k]:bounds[k + 1]], Q,
                         ub[bounds[k]:bounds[k + 1]], ib)
        e.sgd(sh, HYPER, 42, 0, 5, cu.SGD_HOGWILD, True, bounds[k])
        deltas.append(e.Q - Q)
        rates.append(cu.api.item_update_rates(sh))
    tot = rates[0] + rates[1]
    w = [np.where(tot > 0, r / np.maximum(tot, 1e-300), 0.5).astype(np.float32) for r in rates]
    want = Q + sum(wk[:, None] * d for wk, d in zip(w, deltas))
    np.testing.assert_allclose(a["Q"], want, atol=1e-6)
    # an item rated by one rank only keeps that rank's full step
    only0 = (rates[0] > 0) & (rates[1] == 0)
    if only0.any():
        np.testing.assert_allclose(a["Q"][only0], (Q + deltas[0])[only0], atol=1e-6)


def _overlap_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tr, _ = synth.make_ratings(80, 25, 900, min_degree=2, seed=5)
    f = 6
    P, Q, ub, ib = orc.init_model(tr.rows, tr.cols, f)
    bounds = plan_users(tr.rows, world)
    u0, u1 = bounds[rank], bounds[rank + 1]
    eng = OracleEngine(u1 - u0, tr.cols, f, tr.global_bias, P[u0:u1], Q, ub[u0:u1], ib)
    job = ShardedSGD(eng, tr.slice_users(u0, u1), user_offset=u0, sync_every=4, merge="mean", overlap=True)
    assert job.overlap
    job.run(HYPER, 42, 0, 12, cu.SGD_HOGWILD)  # all-reduces started after iterations 4, 8, 12; folded in one period late
    assert job._pending is not None
    job.finish_pending()
    np.savez(os.path.join(out_dir, "o%d.npz" % rank), Q=eng.Q, ib=eng.ib, P=eng.P, exchanges=job.exchanges)
    dist.destroy_process_group()


def test_overlapped_exchange_two_ranks_gloo(tmp_path):
    """overlap=True: the all-reduce of period t is folded in at the end of period t+1; local progress made meanwhile
    is kept on top of the merged base.  Checked against a single-process emulation of exactly that algebra."""
    world, port = 2, 35000 + os.getpid() % 2000
    mp.spawn(_overlap_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r = [np.load(str(tmp_path / ("o%d.npz" % k))) for k in range(world)]
    assert int(r[0]["exchanges"]) == 3
    tr, _ = synth.make_ratings(80, 25, 900, min_degree=2, seed=5)
    f = 6
    P, Q, ub, ib = orc.init_model(tr.rows, tr.cols, f)
    bounds = plan_users(tr.rows, world)
    engs = [OracleEngine(bounds[k + 1] - bounds[k], tr.cols, f, tr.global_bias, P[bounds[k]:bounds[k + 1]], Q,
                         ub[bounds[k]:bounds[k + 1]], ib) for k in range(world)]
    shards = [tr.slice_users(bounds[k], bounds[k + 1]) for k in range(world)]
    base_Q, base_ib = Q.copy(), ib.copy()
    pending = None  # (sum of packed deltas, per-rank snapshots)
    it = 0

    def fold():
        nonlocal base_Q, base_ib, pending
        dQ, dib, snaps = pending
        mQ, mib = base_Q + dQ / np.float32(world), base_ib + dib / np.float32(world)
        for e, (sQ, sib) in zip(engs, snaps):
            e.Q, e.ib = mQ + (e.Q - sQ), mib + (e.ib - sib)
        base_Q, base_ib, pending = mQ, mib, None

    for _ in range(3):
        for k in range(world):
            engs[k].sgd(shards[k], HYPER, 42, it, 4, cu.SGD_HOGWILD, True, bounds[k])
        it += 4
        if pending is not None:
            fold()
        pending = (sum(e.Q - base_Q for e in engs), sum(e.ib - base_ib for e in engs), [(e.Q.copy(), e.ib.copy()) for e in engs])
    fold()
    for k in range(world):
        np.testing.assert_allclose(r[k]["Q"], engs[k].Q, atol=2e-6)
        np.testing.assert_allclose(r[k]["ib"], engs[k].ib, atol=2e-6)
        np.testing.assert_allclose(r[k]["P"], engs[k].P, atol=2e-6)
