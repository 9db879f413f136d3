"""The product's user-sharded driver (cu2rec_amd/csrc/shard_driver.hpp: exchange cadence, wire exchange + merge weights, global
loss reduction, train() over all ranks -- the template cu2rec_shard_job_* / cu2rec_train_sharded / bin/mf -g N instantiate over
HIP) driven on the CPU: tests/host_shard/host_shard.cpp instantiates the SAME template over host memory with the CPU oracle as
the engine, and two gloo ranks run it here (world_size 2, no GPU).  What is compared against: single-process numpy emulations
of the stated merge algebra, and the Python restatement tests/exchange_reference.py.  (The engine is the oracle: these tests
check the DRIVER, the kernels have their own parity tests.)"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import cu2rec_amd as cu
from cu2rec_amd import synth
from cu2rec_amd._lib import Config, Hyper
from conftest import ROOT
from exchange_reference import plan_users
from oracle import oracle as orc

HYPER = (0.01, 0.02, 0.02, 0.02, 0.02)
MERGE = {"mean": 0, "weighted": 1, "sum": 2, "adaptive": 3}
LIB = os.path.join(ROOT, "build", "test", "libcu2rec_shard_host.so")
SRC = os.path.join(ROOT, "tests", "host_shard", "host_shard.cpp")
ALLREDUCE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p)


def build_host_driver():
    """g++ (no HIP): the driver template + the host backend, linked against the oracle's C library."""
    deps = [SRC] + [os.path.join(ROOT, "cu2rec_amd", "csrc", h) for h in ("shard_driver.hpp", "train_schedule_core.hpp", "common.hpp")]
    if os.path.exists(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(d) for d in deps):
        return LIB
    orc.lib()  # builds oracle/_build/libcu2rec_oracle.so if need be
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    obuild = os.path.join(ROOT, "oracle", "_build")
    res = subprocess.run(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-Wall", "-o", LIB, SRC, "-L", obuild, "-lcu2rec_oracle",
                          "-Wl,-rpath," + obuild], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert res.returncode == 0, res.stdout
    return LIB


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def masked(csr, u0, u1):
    """The rank's share of a rating matrix with GLOBAL user ids: rows outside [u0, u1) are empty."""
    ip = csr.indptr.copy()
    ip[:u0 + 1] = csr.indptr[u0]
    ip[u1:] = csr.indptr[u1]
    return np.ascontiguousarray(ip, np.int32)


class HostJob:
    """ctypes face of tests/host_shard/host_shard.cpp: one rank of the C++ driver over host arrays."""

    def __init__(self, rank, world, tr, f, state, u0, u1, sync_every, merge):
        self.L = C.CDLL(build_host_driver())
        self.L.host_shard_last_error.restype = C.c_char_p
        self.P, self.Q, self.ub, self.ib = (np.ascontiguousarray(a, np.float32).copy() for a in state)
        self.tr, self.indptr = tr, masked(tr, u0, u1)
        self.indices, self.data = np.ascontiguousarray(tr.indices, np.int32), np.ascontiguousarray(tr.data, np.float32)

        def allreduce(ctx, buf, count, is_double, stream):
            arr = np.ctypeslib.as_array(C.cast(buf, C.POINTER(C.c_double if is_double else C.c_float)), shape=(count,))
            dist.all_reduce(torch.from_numpy(arr))
            return 0
        self._cb = ALLREDUCE(allreduce)  # (kept alive with the job)
        self.h = C.c_void_p()
        self._ck(self.L.host_shard_create(self._cb, None, rank, world, tr.rows, tr.cols, f, _fp(self.P), _fp(self.Q), _fp(self.ub), _fp(self.ib),
                                          C.c_float(tr.global_bias), _ip(self.indptr), _ip(self.indices), _fp(self.data), int(self.indptr[-1] - self.indptr[0]),
                                          sync_every, MERGE[merge], C.byref(self.h)))

    def _ck(self, rc):
        assert rc == 0, self.L.host_shard_last_error().decode()

    def run(self, hyper, seed, iter0, n, update_items=True):
        self._ck(self.L.host_shard_run(self.h, C.byref(Hyper(*hyper)), C.c_uint64(seed), C.c_uint64(iter0), n, 1 if update_items else 0))

    def exchange(self):
        self._ck(self.L.host_shard_exchange(self.h))

    def loss(self, te, u0, u1):
        ip = masked(te, u0, u1)
        idx, dat = np.ascontiguousarray(te.indices, np.int32), np.ascontiguousarray(te.data, np.float32)
        sa, ss, n, mae, rmse = C.c_double(), C.c_double(), C.c_double(), C.c_float(), C.c_float()
        # (a masked CSR: the rank's own ratings count, rows outside its range are empty and never read)
        self._ck(self.L.host_shard_loss(self.h, _ip(ip), _ip(idx), _fp(dat), te.rows, int(ip[-1] - ip[0]), C.byref(sa), C.byref(ss), C.byref(n),
                                        C.byref(mae), C.byref(rmse)))
        return {"sum_abs": sa.value, "sum_sq": ss.value, "n": n.value, "mae": mae.value, "rmse": rmse.value}

    def info(self):
        se, ex, ut, nt = C.c_int(), C.c_int(), C.c_double(), C.c_double()
        self._ck(self.L.host_shard_info(self.h, C.byref(se), C.byref(ex), C.byref(ut), C.byref(nt)))
        return {"sync_every": se.value, "exchanges": ex.value, "users_total": ut.value, "nnz_total": nt.value}

    def exchange_stats(self):
        n, sec, mx = C.c_int(), C.c_double(), C.c_double()
        self._ck(self.L.host_shard_exchange_stats(self.h, C.byref(n), C.byref(sec), C.byref(mx)))
        return {"timed": n.value, "seconds": sec.value, "max_seconds": mx.value}

    def train(self, te, u0, u1, cfg, verbose=False):
        ip = masked(te, u0, u1)
        idx, dat = np.ascontiguousarray(te.indices, np.int32), np.ascontiguousarray(te.data, np.float32)
        losses = np.full(cfg.total_iterations, np.nan, np.float32)
        self._ck(self.L.host_shard_train(self.h, _ip(ip), _ip(idx), _fp(dat), te.rows, int(ip[-1] - ip[0]), C.byref(cfg), 1 if verbose else 0,
                                         _fp(losses), None))
        return losses

    def close(self):
        self.L.host_shard_destroy(self.h)


def _emulate(tr, f, world, sync, iters, merge):
    """Single process, numpy: `world` logical shards, the stated merge algebra at the stated cadence."""
    P, Q, ub, ib = orc.init_model(tr.rows, tr.cols, f)
    bounds = plan_users(tr.rows, world)
    shards = [orc.CSR(masked(tr, bounds[k], bounds[k + 1]), tr.indices, tr.data, tr.rows, tr.cols) for k in range(world)]
    states = [[a.copy() for a in (P, Q, ub, ib)] for _ in range(world)]
    rates = np.stack([cu.api.item_update_rates(tr.slice_users(bounds[k], bounds[k + 1])) for k in range(world)]).astype(np.float64)
    tot = rates.sum(0)
    if merge == "weighted":
        w = [np.where(tot > 0, r / np.maximum(tot, 1e-300), 1.0 / world).astype(np.float32) for r in rates]
    elif merge == "adaptive":
        # c = 6 scaled with the exchange period: sync / epoch, epoch = nnz / users with ratings (shard_driver.hpp)
        c = 6.0 * min(1.0, sync / max(1.0, tr.nnz / max(np.count_nonzero(np.diff(tr.indptr)), 1)))
        phi = -np.expm1(-c * rates)
        alpha = np.where(phi.sum(0) > 0, -np.expm1(-c * tot) / np.maximum(phi.sum(0), 1e-300), 1.0).astype(np.float32)
        w = [alpha] * world
    else:
        w = [None] * world
    scale = np.float32(1.0 / world) if merge == "mean" else np.float32(1.0)
    Qb, ibb = Q.copy(), ib.copy()
    it, exchanges = 0, 0
    while it < iters:
        n = min(sync, iters - it)
        for k in range(world):
            orc.sgd_iterations(shards[k], *states[k], tr.global_bias, HYPER, 42, it, n, dot_order=orc.DOT_TREE16)
        it += n
        if n == sync:
            dQ = sum((s[1] - Qb) * (wk[:, None] if wk is not None else np.float32(1)) for s, wk in zip(states, w))
            dib = sum((s[3] - ibb) * (wk if wk is not None else np.float32(1)) for s, wk in zip(states, w))
            Qb, ibb = Qb + scale * dQ, ibb + scale * dib
            for s in states:
                s[1][...] = Qb
                s[3][...] = ibb
            exchanges += 1
    return states, bounds, exchanges


def _merge_worker(rank, world, port, out_dir, merge):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tr, te = synth.make_ratings(80, 25, 900, min_degree=2, seed=5)
    f = 6
    bounds = plan_users(tr.rows, world)
    u0, u1 = bounds[rank], bounds[rank + 1]
    job = HostJob(rank, world, tr, f, orc.init_model(tr.rows, tr.cols, f), u0, u1, 4, merge)
    job.run(HYPER, 42, 0, 10)  # exchanges behind iterations 4 and 8, then two local iterations
    res = job.loss(te, u0, u1)
    info = job.info()
    xs = job.exchange_stats()  # (the driver's exchange timers: one completed event pair per exchange, a ring of 32)
    np.savez(os.path.join(out_dir, "m%d.npz" % rank), P=job.P, Q=job.Q, ub=job.ub, ib=job.ib, exchanges=info["exchanges"], rmse=res["rmse"],
             n=res["n"], users_total=info["users_total"], timed=xs["timed"], seconds=xs["seconds"], max_seconds=xs["max_seconds"])
    job.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("merge", ["mean", "weighted", "adaptive", "sum"])
def test_cpp_driver_two_ranks_gloo_against_the_merge_algebra(tmp_path, merge):
    world, port = 2, 36000 + os.getpid() % 2000 + 7 * MERGE[merge]
    mp.spawn(_merge_worker, args=(world, port, str(tmp_path), merge), nprocs=world, join=True)
    r = [np.load(str(tmp_path / ("m%d.npz" % k))) for k in range(world)]
    tr, te = synth.make_ratings(80, 25, 900, min_degree=2, seed=5)
    states, bounds, exchanges = _emulate(tr, 6, world, 4, 10, merge)
    assert int(r[0]["exchanges"]) == int(r[1]["exchanges"]) == exchanges == 2
    for k in range(world):  # every exchange timed, on every rank
        assert int(r[k]["timed"]) == 2 and 0 < float(r[k]["max_seconds"]) <= float(r[k]["seconds"]) <= 60.0
    assert float(r[0]["users_total"]) == float(np.count_nonzero(np.diff(tr.indptr)))  # the constructor's all-reduce of the totals
    for k in range(world):
        u0, u1 = bounds[k], bounds[k + 1]
        np.testing.assert_allclose(r[k]["P"][u0:u1], states[k][0][u0:u1], atol=1e-6)
        np.testing.assert_allclose(r[k]["Q"], states[k][1], atol=1e-6)
        np.testing.assert_allclose(r[k]["ub"][u0:u1], states[k][2][u0:u1], atol=1e-6)
        np.testing.assert_allclose(r[k]["ib"], states[k][3], atol=1e-6)
    assert float(r[0]["rmse"]) == float(r[1]["rmse"]) and int(r[0]["n"]) == te.nnz  # the loss is reduced over the ranks


def test_cpp_driver_one_rank_is_the_oracle_bit_for_bit_and_frozen_items_terminate():
    """World 1 through the C++ driver: no exchange ever, the result is the engine's; and with update_items == 0 a run longer
    than a period terminates (the period counter starts over although nothing is exchanged)."""
    dist.init_process_group("gloo", rank=0, world_size=1, init_method="tcp://127.0.0.1:%d" % (38000 + os.getpid() % 2000))
    try:
        tr, te = synth.make_ratings(60, 30, 600, min_degree=2, seed=2)
        f = 8
        job = HostJob(0, 1, tr, f, orc.init_model(tr.rows, tr.cols, f), 0, tr.rows, 3, "adaptive")
        job.run(HYPER, 42, 0, 10)
        want = orc.init_model(tr.rows, tr.cols, f)
        orc.sgd_iterations(orc.CSR(tr.indptr, tr.indices, tr.data, tr.rows, tr.cols), *want, tr.global_bias, HYPER, 42, 0, 10, dot_order=orc.DOT_TREE16)
        for g, w in zip((job.P, job.Q, job.ub, job.ib), want):
            np.testing.assert_array_equal(g, w)
        assert job.info()["exchanges"] == 0 and job.loss(te, 0, te.rows)["n"] == te.nnz
        q_before = job.Q.copy()
        job.run(HYPER, 42, 10, 25, update_items=False)  # 8 periods of 3 + 1
        np.testing.assert_array_equal(job.Q, q_before)
        assert job.info()["exchanges"] == 0
        job.close()
    finally:
        dist.destroy_process_group()


def _oracle_factory(rows, cols, f, gb, P0, ub0, tr, te):
    from test_parallel_cpu import OracleEngine
    Q = orc.normal_fill(cols * f, f).reshape(cols, f)
    ib = orc.normal_fill(cols, f)
    return OracleEngine(rows, cols, f, gb, P0, Q, ub0, ib), tr, te


def _train_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tr, te = synth.make_ratings(90, 25, 1000, min_degree=2, seed=6)
    f = 5
    kw = dict(total_iterations=12, n_factors=f, check_error=4, learning_rate=0.02)
    bounds = plan_users(tr.rows, world)
    u0, u1 = bounds[rank], bounds[rank + 1]
    cfg = cu.default_config(**kw)
    job = HostJob(rank, world, tr, f, orc.init_model(tr.rows, tr.cols, f), u0, u1, 3, "mean")
    losses = job.train(te, u0, u1, cfg)
    # the Python restatement of the same driver (tests/exchange_reference.py), same ranks, same engine arithmetic up to the dot order
    from exchange_reference import train_sharded
    cfg_py = cu.default_config(**kw)
    _, Qp, losses_py, _, ibp, _ = train_sharded(tr, te, cfg_py, sync_every=3, merge="mean", engine_factory=_oracle_factory, verbose=False)
    np.savez(os.path.join(out_dir, "t%d.npz" % rank), Q=job.Q, ib=job.ib, losses=losses, lr=cfg.learning_rate, cur=cfg.cur_iterations,
             exchanges=job.info()["exchanges"], Qp=Qp, ibp=ibp, losses_py=losses_py, lr_py=cfg_py.learning_rate)
    job.close()
    dist.destroy_process_group()


def test_cpp_driver_train_sharded_two_ranks_gloo(tmp_path):
    """shard_train (cu2rec_train_sharded's loop) on two ranks: both ranks hold the same item side behind the final exchange, saw the
    same GLOBAL losses at the reference's check iterations and took the same learning-rate decisions; and the run agrees with the
    Python restatement of the driver."""
    world, port = 2, 39000 + os.getpid() % 2000
    mp.spawn(_train_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    a, b = (np.load(str(tmp_path / ("t%d.npz" % k))) for k in range(2))
    np.testing.assert_array_equal(a["Q"], b["Q"])
    np.testing.assert_array_equal(a["ib"], b["ib"])
    np.testing.assert_array_equal(np.nan_to_num(a["losses"]), np.nan_to_num(b["losses"]))
    assert float(a["lr"]) == float(b["lr"]) and int(a["cur"]) == 12
    checked = [i for i in range(12) if not np.isnan(a["losses"][i])]
    assert checked == [0, 3, 7, 11] and a["losses"][11] < a["losses"][0]
    # a period of 3 that starts over at every exchange: one out of cadence in front of the check at iteration 1; iterations 2-4: one in
    # cadence (the check at 4 falls on it); iterations 5-8 and 9-12: one in cadence after three iterations, one out of cadence in front
    # of the check behind the fourth
    assert int(a["exchanges"]) == 6
    np.testing.assert_allclose(a["Q"], a["Qp"], atol=2e-6)
    np.testing.assert_allclose(a["ib"], a["ibp"], atol=2e-6)
    np.testing.assert_allclose(np.nan_to_num(a["losses"]), np.nan_to_num(a["losses_py"]), atol=2e-6)
    assert abs(float(a["lr"]) - float(a["lr_py"])) < 1e-9
