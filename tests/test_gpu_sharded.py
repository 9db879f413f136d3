"""The C++ multi-GPU driver (cu2rec_amd/csrc/sharded.cpp) on ONE GPU (-m gpu): world 1, RCCL at world 1, and two ranks
sharing the GPU with the callback communicator (gloo through the host) -- against the CPU oracle running the same
schedule with numpy doing the merge.  Real N = 2 / 4 / 8 runs are the driver's scaling bench."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

import cu2rec_amd as cu
from cu2rec_amd import sharded, synth
from conftest import ROOT
from oracle import oracle as orc

pytestmark = pytest.mark.gpu

HYPER = (0.01, 0.02, 0.02, 0.02, 0.02)


def _as_orc(m):
    return orc.CSR(m.indptr, m.indices, m.data, m.rows, m.cols, m.global_bias)


def _masked(tr, u0, u1):
    ip = tr.indptr.copy()
    ip[:u0 + 1] = tr.indptr[u0]
    ip[u1:] = tr.indptr[u1]
    return orc.CSR(ip, tr.indices, tr.data, tr.rows, tr.cols)


def test_world1_train_sharded_equals_train():
    """One rank: cu2rec_train_sharded is cu2rec_train -- every output and every logged loss bit for bit."""
    tr, te = synth.make_ratings(600, 80, 9000, min_degree=3, seed=6)
    kw = dict(total_iterations=45, n_factors=12, check_error=15, learning_rate=0.02)
    cfg1 = cu.default_config(**kw)
    want = cu.train(tr, te, cfg1, mode="ordered", verbose=False)
    cfg2 = cu.default_config(**kw)
    comm = sharded.Comm(0, 1)
    P, Q, losses, ub, ib, (u0, u1), stats = sharded.train_sharded(comm, tr, te, cfg2, mode="ordered", sync_every=7, verbose=False)
    assert (u0, u1) == (0, tr.rows) and stats.n_checks == 4
    for g, w in zip((P, Q, losses, ub, ib), want):
        np.testing.assert_array_equal(g, w)
    assert cfg1.learning_rate == cfg2.learning_rate and cfg1.cur_iterations == cfg2.cur_iterations


def test_rccl_world1_allreduce_in_the_loop(monkeypatch):
    """ncclAllReduce really runs (a one-rank communicator, CU2REC_RCCL_WORLD1=1): pack -> all-reduce -> apply every 5
    iterations is the identity up to the rounding of Q_base + (Q - Q_base), and the loss sums come back unchanged."""
    monkeypatch.setenv("CU2REC_RCCL_WORLD1", "1")
    tr, te = synth.make_ratings(2000, 150, 30000, min_degree=3, seed=9)
    f = 40
    comm = sharded.Comm(0, 1)
    model = cu.Model(tr.rows, tr.cols, f, tr.global_bias)
    d_tr, d_te = cu.DeviceCSR(tr), cu.DeviceCSR(te)
    job = sharded.ShardJob(comm, model, d_tr, sync_every=5, merge="sum")
    job.run(HYPER, 42, 0, 23, mode="ordered")
    info = job.info()
    assert info["exchanges"] == 4 and info["wire_bytes"] == tr.cols * (f + 1) * 4
    assert info["users_total"] == np.count_nonzero(np.diff(tr.indptr)) and info["nnz_total"] == tr.nnz
    P, Q, ub, ib = orc.init_model(tr.rows, tr.cols, f)
    orc.sgd_iterations(_as_orc(tr), P, Q, ub, ib, tr.global_bias, HYPER, 42, 0, 23, dot_order=orc.DOT_TREE16)
    for g, w in zip(model.download(), (P, Q, ub, ib)):
        assert float(np.abs(g.astype(np.float64) - w).max()) <= 2e-6
    got, plain = job.loss(d_te), model.loss(d_te)
    assert got["n"] == te.nnz and got["sum_sq"] == plain["sum_sq"] and got["rmse"] == plain["rmse"]
    job.close()
    comm.close()


# ---- two ranks on one GPU, the all-reduce through gloo ---------------------------------------------------------------

def _two_rank_worker(rank, world, port, out_dir, merge):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import cu2rec_amd as cu
    from cu2rec_amd import sharded, synth
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cu.lib().cu2rec_hogwild_resident(0)  # the ranks share a GPU: a resident launch wants it to itself
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipDeviceSynchronize.argtypes = []

    def allreduce(ctx, buf, count, is_double, stream):  # the product calls back with a device buffer
        try:
            host = np.empty(count, np.float64 if is_double else np.float32)
            if hip.hipDeviceSynchronize() != 0 or hip.hipMemcpy(host.ctypes.data, buf, host.nbytes, 2) != 0:
                return 1
            t = torch.from_numpy(host)
            dist.all_reduce(t)
            return 0 if hip.hipMemcpy(buf, host.ctypes.data, host.nbytes, 1) == 0 else 1
        except Exception:
            return 1

    tr, te = synth.make_ratings(500, 150, 12000, min_degree=3, seed=21)
    cfg = cu.default_config(total_iterations=18, n_factors=24, check_error=9, learning_rate=0.01)
    comm = sharded.Comm(rank, world, allreduce=allreduce)
    P, Q, losses, ub, ib, (u0, u1), stats = sharded.train_sharded(comm, tr, te, cfg, mode="ordered", sync_every=3, merge=merge,
                                                                  verbose=False)
    np.savez(os.path.join(out_dir, "r%d.npz" % rank), P=P, Q=Q, ub=ub, ib=ib, losses=losses, u0=u0, u1=u1, lr=cfg.learning_rate,
             updates=stats.updates)
    comm.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("merge", ["mean", "weighted", "adaptive"])
def test_two_ranks_one_gpu_cpp_driver_matches_oracle(tmp_path, merge):
    """cu2rec_train_sharded on two ranks (ordered mode: deterministic per shard), exchanges every 3 iterations, against the
    oracle running both shards and merging with numpy: P slices and user biases bit for bit, the item side within one
    ulp-sized rounding of the merge (2e-7), identical replicas, identical logged losses on both ranks."""
    import torch.multiprocessing as mp
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_two_rank_worker, args=(2, port, str(tmp_path), merge), nprocs=2, join=True)
    r = [np.load(os.path.join(str(tmp_path), "r%d.npz" % k)) for k in range(2)]
    tr, te = synth.make_ratings(500, 150, 12000, min_degree=3, seed=21)
    f, sync, total = 24, 3, 18
    bounds = [(int(x["u0"]), int(x["u1"])) for x in r]
    assert bounds[0][1] == bounds[1][0] and bounds[1][1] == tr.rows
    P, Q, ub, ib = orc.init_model(tr.rows, tr.cols, f)
    oP, oub = [P.copy() for _ in bounds], [ub.copy() for _ in bounds]
    oQ, oib = [Q.copy() for _ in bounds], [ib.copy() for _ in bounds]
    Qb, ibb = Q.copy(), ib.copy()
    masked = [_masked(tr, u0, u1) for u0, u1 in bounds]
    if merge == "weighted":
        rates = [cu.api.item_update_rates(tr.slice_users(u0, u1)) for u0, u1 in bounds]
        tot = rates[0] + rates[1]
        w = [np.where(tot > 0, rk / np.where(tot > 0, tot, 1), 0.5).astype(np.float32) for rk in rates]
        scale = np.float32(1.0)
    elif merge == "adaptive":  # the sum, scaled per item by phi(r_total) / sum_k phi(r_k), phi(r) = 1 - exp(-6 r)
        rates = [cu.api.item_update_rates(tr.slice_users(u0, u1)).astype(np.float64) for u0, u1 in bounds]
        phi = lambda r: -np.expm1(-6.0 * r)
        den = phi(rates[0]) + phi(rates[1])
        alpha = np.where(den > 0, phi(rates[0] + rates[1]) / np.where(den > 0, den, 1), 1.0).astype(np.float32)
        w = [alpha, alpha]
        scale = np.float32(1.0)
    else:
        w = [np.ones(tr.cols, np.float32)] * 2
        scale = np.float32(0.5)
    def merge_now():
        nonlocal Qb, ibb
        dQ = w[0][:, None] * (oQ[0] - Qb) + w[1][:, None] * (oQ[1] - Qb)
        dib = w[0] * (oib[0] - ibb) + w[1] * (oib[1] - ibb)
        Qb, ibb = Qb + scale * dQ, ibb + scale * dib
        for k in range(2):
            oQ[k], oib[k] = Qb.copy(), ibb.copy()

    # the driver's cadence: an exchange every `sync` iterations, and one out of cadence in front of every loss check
    # (i == 0, every check_error = 9, last) so that the loss is taken on reconciled item factors
    since, it = 0, 0
    for seg_end in (0, 8, 17):
        while it <= seg_end:
            n = min(seg_end + 1 - it, sync - since)
            for k in range(2):
                orc.sgd_iterations(masked[k], oP[k], oQ[k], oub[k], oib[k], tr.global_bias, HYPER, 42, it, n, dot_order=orc.DOT_TREE16)
            it += n
            since += n
            if since >= sync:
                merge_now()
                since = 0
        if since > 0:
            merge_now()
            since = 0
    for k, (u0, u1) in enumerate(bounds):
        # the user side depends on the item side it saw, which carries the merge's rounding: tolerance, not bits
        assert float(np.abs(r[k]["P"] - oP[k][u0:u1]).max()) <= 2e-6
        assert float(np.abs(r[k]["ub"] - oub[k][u0:u1]).max()) <= 2e-6
        assert float(np.abs(r[k]["Q"] - Qb).max()) <= 2e-6 and float(np.abs(r[k]["ib"] - ibb).max()) <= 2e-6
    np.testing.assert_array_equal(r[0]["Q"], r[1]["Q"])  # identical replicas
    np.testing.assert_array_equal(r[0]["ib"], r[1]["ib"])
    np.testing.assert_array_equal(r[0]["losses"], r[1]["losses"])  # the GLOBAL test RMSE on both ranks
    assert np.isfinite(r[0]["losses"][[0, 8, 17]]).all() and float(r[0]["lr"]) == float(r[1]["lr"])
    assert float(r[0]["updates"]) == float(np.count_nonzero(np.diff(tr.indptr))) * total


def test_bin_mf_g1_flag_and_sharded_cli_single_rank(tmp_path):
    """bin/mf accepts the multi-GPU flags; with -g 1 it is the single-GPU program (same files as without the flags)."""
    import subprocess
    tr, te = synth.make_ratings(300, 60, 4000, min_degree=3, seed=3)
    outs = []
    for k, extra in enumerate(([], ["-g", "1", "-s", "5", "-w", "mean"])):
        d = tmp_path / ("run%d" % k)
        d.mkdir()
        synth.write_csv(str(d / "train.csv"), tr)
        synth.write_csv(str(d / "test.csv"), te)
        (d / "c.cfg").write_text("0 20 8 0.01 42 0.02 0.02 0.02 0.02\n")
        res = subprocess.run([os.path.join(ROOT, "bin", "mf"), "-c", str(d / "c.cfg"), "-m", "ordered"] + extra +
                             [str(d / "train.csv"), str(d / "test.csv")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                             timeout=300)
        assert res.returncode == 0, res.stderr
        outs.append({c: (d / ("train_f8_%s.csv" % c)).read_text() for c in ("p", "q", "user_bias", "item_bias", "global_bias")})
    assert outs[0] == outs[1]


def _emulated_shards_gap(workload, f, iters, sync, n, merges=("adaptive",)):
    """N user shards emulated on ONE GPU (N engines run one after the other, block-solve mode per shard, the merge arithmetic of the
    driver's exchange in torch): test RMSE after `iters` iterations minus the unsharded = sequential result, per merge rule.  The
    driver's own exchange (wire pack / ncclAllReduce / apply) is pinned against the oracle by the two-rank test above; this pins
    the CONVERGENCE of the merge rule at BASELINE.json configs[3] / [4]'s shapes."""
    import torch
    import bench
    from cu2rec_amd.engine import DeviceRatings, Engine
    from cu2rec_amd.sharded import plan_users
    train, test = bench.load_dataset(workload, 20240917, 0, lambda: None)
    dev = torch.device("cuda", 0)

    def test_rmse(engines, bounds):
        ss = 0.0
        for e, (u0, u1) in zip(engines, bounds):
            ss += e.loss(DeviceRatings(test.slice_users(u0, u1), dev))["sum_sq"]
        return (ss / test.nnz) ** 0.5

    base = Engine(train.rows, train.cols, f, train.global_bias, device=dev)
    base.sgd(DeviceRatings(train, dev), HYPER, 42, 0, iters, "blocksolve")
    ref = test_rmse([base], [(0, train.rows)])
    del base
    P0 = cu.initialize_normal_array(train.rows * f, f).reshape(train.rows, f)
    ub0 = cu.initialize_normal_array(train.rows, f)
    b = plan_users(train.rows, n)
    bounds = list(zip(b[:-1], b[1:]))
    shards = [train.slice_users(u0, u1) for u0, u1 in bounds]
    rates = np.stack([cu.api.item_update_rates(s) for s in shards]).astype(np.float64)
    tot = rates.sum(0)
    phi = lambda x: -np.expm1(-6.0 * x)
    alpha = torch.tensor(np.where(tot > 0, phi(tot) / np.maximum(phi(rates).sum(0), 1e-300), 1.0), dtype=torch.float32, device=dev)
    weights = torch.tensor(np.where(tot > 0, rates / np.maximum(tot, 1e-300), 1.0 / n), dtype=torch.float32, device=dev)
    got = {}
    for merge in merges:
        engines = [Engine(u1 - u0, train.cols, f, train.global_bias, P=P0[u0:u1], user_bias=ub0[u0:u1], device=dev) for u0, u1 in bounds]
        d = [DeviceRatings(s, dev) for s in shards]
        Qb, ibb = engines[0].Q.clone(), engines[0].item_bias.clone()
        it = 0
        while it < iters:
            k = min(sync, iters - it)
            for e, dr, (u0, _) in zip(engines, d, bounds):
                e.sgd(dr, HYPER, 42, it, k, "blocksolve", True, u0)
            it += k
            dQ = torch.stack([e.Q - Qb for e in engines])[:, :train.cols]
            dib = torch.stack([e.item_bias - ibb for e in engines])[:, :train.cols]
            if merge == "adaptive":
                Qb[:train.cols] += alpha[:, None] * dQ.sum(0)
                ibb[:train.cols] += alpha * dib.sum(0)
            else:
                Qb[:train.cols] += (weights[:, :, None] * dQ).sum(0)
                ibb[:train.cols] += (weights * dib).sum(0)
            for e in engines:
                e.Q.copy_(Qb)
                e.item_bias.copy_(ibb)
        got[merge] = test_rmse(engines, bounds) - ref
        del engines, d
    return got


def test_eight_shards_adaptive_merge_within_the_stated_tolerance_of_the_sequential_run():
    """The acceptance tolerance of a user-sharded run: after 1,000 iterations of the ML-20M shape, f=100, the
    test RMSE of N = 8 shards -- block-solve mode per shard, item deltas reconciled once per epoch (115 iterations) with the
    driver's `adaptive` merge, delta = phi(r_total) / sum_k phi(r_k) * sum_k delta_k, phi(r) = 1 - exp(-6 r) -- stays within
    6e-4 of the unsharded = sequential result (measured in round 2: +3.9e-4; `weighted`: +1.07e-3, pinned here as the bar the
    adaptive merge must beat)."""
    got = _emulated_shards_gap("ml-20m", 100, 1000, 115, 8, ("adaptive", "weighted"))
    assert abs(got["adaptive"]) <= 6e-4, got
    assert abs(got["adaptive"]) < abs(got["weighted"]), got


def test_two_shards_adaptive_merge_meets_the_1e4_bar():
    """BASELINE.json configs[3] at N = 2 (emulated): the sharded run's test RMSE after 1,000 iterations stays within the north
    star's 1e-4 of the sequential result (measured: -1e-5)."""
    got = _emulated_shards_gap("ml-20m", 100, 1000, 115, 2)
    assert abs(got["adaptive"]) <= 1e-4, got


def test_four_shards_adaptive_merge_is_outside_the_1e4_bar_by_this_much():
    """BASELINE.json configs[3] at N = 4 (emulated): measured +1.7e-4 -- OUTSIDE the north star's 1e-4 (a sharded run reconciles the
    item side once per epoch; it is not the sequential run), inside the stated tolerance of 3e-4 for N = 4.  No merge rule tried
    brings N >= 4 under 1e-4 (exchanging more often does not help, the bias is in the merge)."""
    got = _emulated_shards_gap("ml-20m", 100, 1000, 115, 4)
    assert abs(got["adaptive"]) <= 3e-4, got


def test_eight_shards_netflix_f128_within_the_stated_tolerance():
    """BASELINE.json configs[4] (Netflix shape, f=128, 8 shards, emulated): 660 iterations, exchange every 165 (one epoch): the
    sharded run's test RMSE sits 1.2e-3 BELOW the sequential run's at that point of the trajectory (nearly every one of the 17,770
    items is updated many times per iteration by every shard: the adaptive merge is the mean there).  Pinned: |gap| <= 1.6e-3, and
    the sign."""
    got = _emulated_shards_gap("netflix", 128, 660, 165, 8)
    assert -1.6e-3 <= got["adaptive"] <= 0.0, got


def test_bench_self_launches_its_ranks(tmp_path):
    """`python bench.py --gpus 2` with NO launcher around it: the program starts its two ranks itself (torch.distributed.run, fresh
    processes; here sharing the one GPU through gloo) and rank 0's JSON line says n_gpus 2."""
    import json
    import subprocess
    env = dict(os.environ, CU2REC_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--workload", "ml-1m",
                          "--factors", "50", "--no-cpu-baseline", "--no-side-modes"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 4 and line["value"] > 0
    assert line["config"]["sharded_run_tolerance"] is not None
