#!/usr/bin/env python3
"""Benchmark of the cu2rec hot path on MI355X: SGD updates/sec (+ test RMSE), ML-20M shape, f=100.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one reference iteration: one SGD update for every user with at least one rating
(sgd.cu:27-37).  The library is called with 500 iterations at a time, the reference's stretch between two
loss checks (N > 1: one exchange period);
such a call is ONE resident launch (user rows in the register file, a grid barrier between
iterations: cu2rec_amd/csrc/resident.hip) when the rows fit, else one launch per iteration.
Inputs (CSR, P, Q, biases) are resident in HBM before the timed region.  For N > 1 every rank holds its own
ML-20M-sized user population (weak scaling; same item set) and the replicas of Q / item_bias are
reconciled by one RCCL all-reduce every `--sync-every` steps (default: one epoch = nnz / users
steps), inside the timed region.

Prints ONE JSON line (rank 0).  Extra objects:
  roofline     HBM roofline of the SGD kernel: algorithmic bytes per launch (16 f + 32 per update,
               SURVEY.md section 8d) / average kernel duration measured here with HIP events.
  cpu_baseline the reference's own CPU twin (oracle/_ref/mf_cpu, kind "reference") timed on a
               bounded sample of the same workload on this box's host cores, plus the oracle port
               with the counter-based sampler ("port").
"""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)


def load_dataset(name, seed, rank, barrier):
    """Synthetic set with the named shape; generated once per box and cached under /tmp."""
    from cu2rec_amd import synth
    from cu2rec_amd.api import HostCSR
    path = os.path.join(tempfile.gettempdir(), "cu2rec_synth_%s_%d.npz" % (name, seed))
    if rank == 0 and not os.path.exists(path):
        tr, te = synth.make_named(name, seed=seed)
        tmp = path + ".tmp.npz"
        np.savez(tmp, tr_indptr=tr.indptr, tr_indices=tr.indices, tr_data=tr.data, te_indptr=te.indptr,
                 te_indices=te.indices, te_data=te.data, shape=np.array([tr.rows, tr.cols]),
                 gb=np.array([tr.global_bias], np.float32))
        os.replace(tmp, path)
    barrier()
    z = np.load(path)
    rows, cols = (int(v) for v in z["shape"])
    gb = float(z["gb"][0])
    return (HostCSR(z["tr_indptr"], z["tr_indices"], z["tr_data"], rows, cols, gb),
            HostCSR(z["te_indptr"], z["te_indices"], z["te_data"], rows, cols, gb))


def cpu_baseline(train, test, f, hyper, budget_s=12.0):
    """Reference CPU twin (if its binary travelled) and the oracle port, on bounded samples."""
    from cu2rec_amd import synth
    from oracle import oracle as orc
    out = {}
    # ---- oracle port, counter-based sampler, 1 thread: whole training set, a few iterations
    o_tr = orc.CSR(train.indptr, train.indices, train.data, train.rows, train.cols, train.global_bias)
    P, Q, ub, ib = orc.init_model(train.rows, train.cols, f)
    users = int(np.count_nonzero(np.diff(train.indptr)))
    t0 = time.perf_counter()
    orc.sgd_iterations(o_tr, P, Q, ub, ib, train.global_bias, hyper, 42, 0, 1)
    one = time.perf_counter() - t0
    iters = int(max(1, min(200, (budget_s / 2) / max(one, 1e-6))))
    t0 = time.perf_counter()
    orc.sgd_iterations(o_tr, P, Q, ub, ib, train.global_bias, hyper, 42, 1, iters)
    dt = time.perf_counter() - t0
    port = {"value": users * iters / dt, "unit": "updates/s", "cores": 1, "kind": "port",
            "sample": "oracle/cu2rec_oracle.c (mf_sequential.cu:102-143 with the Philox sampler), full training set, "
                      "%d iterations, f=%d, gcc -O3 -ffp-contract=off" % (iters, f)}
    # ---- the same port, Hogwild over users on all host cores (OpenMP) -- the strongest CPU form of this path
    P, Q, ub, ib = orc.init_model(train.rows, train.cols, f)
    threads = orc.sgd_iterations_parallel(o_tr, P, Q, ub, ib, train.global_bias, hyper, 42, 0, 2)
    t0 = time.perf_counter()
    it_par = 0
    while time.perf_counter() - t0 < budget_s / 4:
        orc.sgd_iterations_parallel(o_tr, P, Q, ub, ib, train.global_bias, hyper, 42, 2 + it_par, 10)
        it_par += 10
    dt = time.perf_counter() - t0
    port["all_cores"] = {"value": users * it_par / dt, "unit": "updates/s", "cores": threads, "kind": "port",
                         "sample": "same port, users of an iteration split over %d OpenMP threads racing on the item rows "
                                   "(Hogwild on the host), %d iterations" % (threads, it_par)}
    # ---- the reference's own binary on a user subsample written as CSV
    exe = orc.ref_binary("mf_cpu")
    if exe:
        n_users = min(train.rows, 6000)
        sub_tr, sub_te = train.slice_users(0, n_users), test.slice_users(0, n_users)
        iters_ref = 30
        with tempfile.TemporaryDirectory() as td:
            ptr, pte, pcfg = os.path.join(td, "tr.csv"), os.path.join(td, "te.csv"), os.path.join(td, "c.cfg")
            synth.write_csv(ptr, sub_tr)
            synth.write_csv(pte, sub_te)
            with open(pcfg, "w") as fh:
                fh.write("0 %d %d %g 42 %g %g %g %g\n" % ((iters_ref, f) + tuple(hyper)))
            res = subprocess.run([exe, "-c", pcfg, ptr, pte], stdout=subprocess.PIPE, text=True, timeout=600)
        m = re.search(r"Time taken for (\d+) of iterations is ([0-9.]+)", res.stdout)
        if res.returncode == 0 and m and float(m.group(2)) > 0:
            users_sub = int(np.count_nonzero(np.diff(sub_tr.indptr)))
            out = {"value": users_sub * iters_ref / float(m.group(2)), "unit": "updates/s", "cores": 1,
                   "kind": "reference",
                   "sample": "nickgreenquist/cu2rec mf_sequential.cu compiled unmodified (oracle/_ref/mf_cpu): first %d "
                             "users (%d ratings) of the workload, %d iterations, f=%d; its own clock() timer, which "
                             "includes its two loss evaluations; ~97%% of its time is per-update std::random_device + "
                             "mt19937 construction (mf_sequential.cu:109-110)" % (n_users, sub_tr.nnz, iters_ref, f),
                   "port": port}
    if not out:
        out = port
    return out


def measured_traffic(workload, f, name):
    """HBM bytes per launch of the SGD kernel from the committed rocprofv3 PMC passes (profiles/), priced as
    MI355X_MICROARCH.md prescribes (separate FETCH_SIZE / WRITE_SIZE passes, KiB units, FETCH_SIZE doubled on
    gfx950).  Counters cannot be read from inside this process, so the figure is the latest committed one for
    this exact workload / kernel; None if there is none."""
    import glob
    tag = "%s_f%d" % (workload.replace("-", ""), f)
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_%s.json" % tag)), reverse=True):
        try:
            for kernel, k in json.load(open(path)).items():
                if name in kernel and "hbm_bytes_per_launch_corrected" in k:
                    return {"bytes_per_launch": k["hbm_bytes_per_launch_corrected"],
                            "source": os.path.relpath(path, ROOT) + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate "
                                      "passes; raw %.1f MB, FETCH_SIZE x2 gfx950 correction applied)"
                                      % (k["hbm_bytes_per_launch_raw"] / 1e6)}
        except (OSError, ValueError):
            continue
    return None


def log(msg):
    sys.stderr.write("[bench %7.1fs] %s\n" % (time.perf_counter() - T_START, msg))
    sys.stderr.flush()


T_START = time.perf_counter()


def main():
    import faulthandler
    faulthandler.enable()
    if os.environ.get("CU2REC_BENCH_WATCHDOG"):
        faulthandler.dump_traceback_later(int(os.environ["CU2REC_BENCH_WATCHDOG"]), repeat=True)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=500)
    ap.add_argument("--workload", default="ml-20m", choices=["ml-100k", "ml-1m", "ml-20m", "ml-25m", "netflix"])
    ap.add_argument("--factors", type=int, default=100)
    ap.add_argument("--mode", default="hogwild", choices=["hogwild", "serial", "ordered"])
    ap.add_argument("--sync-every", type=int, default=0, help="steps between item-factor all-reduces (0 = one epoch)")
    ap.add_argument("--merge", default="mean", choices=["mean", "sum", "weighted"])
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: every rank holds its own full-size user population (default); strong: ONE dataset, "
                         "users sharded across the ranks (BASELINE.json configs[3])")
    ap.add_argument("--seed", type=int, default=20240917)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--kernel-samples", type=int, default=200)
    ap.add_argument("--iters-per-launch", type=int, default=1,
                    help="Hogwild launch blocking (opt-in, not the reference cadence): updates per user per launch")
    ap.add_argument("--ordered-steps", type=int, default=128,
                    help="extra (untimed for `value`) pass in the exact ordered mode, reported beside the headline; 0 = skip")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the cu2rec_amd hot path has no CPU fallback")
    # CU2REC_BENCH_BACKEND=gloo lets several ranks share one GPU (plumbing check on a 1-GPU box); the real
    # multi-GPU run is backend nccl (= RCCL over xGMI), one rank per GPU
    backend = os.environ.get("CU2REC_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    def barrier():
        if world > 1:
            if backend == "nccl":
                dist.barrier(device_ids=[dev_index])
            else:
                dist.barrier()

    import cu2rec_amd as cu
    from cu2rec_amd.engine import DeviceRatings, Engine
    from cu2rec_amd.parallel import ShardedSGD

    log("torch + library loaded, device %s" % torch.cuda.get_device_name(device))
    if world > 1 and backend != "nccl":
        cu.lib().cu2rec_hogwild_resident(0)  # ranks share a GPU here: a resident launch needs the GPU to itself
    train, test = load_dataset(args.workload, args.seed, rank, barrier)
    log("dataset ready: %d users, %d items, %d train ratings" % (train.rows, train.cols, train.nnz))
    f = args.factors
    hyper = (0.01, 0.02, 0.02, 0.02, 0.02)  # preprocessing/create_config.py:25-32
    # iterations per cu2rec_sgd_update call: N > 1 -> the exchange period (default one epoch = nnz / users iterations);
    # N = 1 -> nothing to exchange: 500 per call, the stretch between two loss checks of the reference's loop
    # (check_error = 500, config.h:41-51, training.cu:118) and what cu2rec_train issues
    if world > 1:
        sync_every = args.sync_every or max(1, int(round(train.nnz / max(int(np.count_nonzero(np.diff(train.indptr))), 1))))
    else:
        sync_every = args.sync_every or 500
    user_offset = rank * train.rows  # weak scaling: rank r's users are users [r * rows, (r+1) * rows) of the population
    if args.scaling == "strong" and world > 1:
        from cu2rec_amd.parallel import plan_users
        bounds = plan_users(train.rows, world)
        user_offset = bounds[rank]
        train, test = train.slice_users(bounds[rank], bounds[rank + 1]), test.slice_users(bounds[rank], bounds[rank + 1])
    users_active = int(np.count_nonzero(np.diff(train.indptr)))

    eng = Engine(train.rows, train.cols, f, train.global_bias, device=device)
    d_train, d_test = DeviceRatings(train, device), DeviceRatings(test, device)
    rates = cu.api.item_update_rates(train) if args.merge == "weighted" else None
    job = ShardedSGD(eng, d_train, user_offset=user_offset, sync_every=sync_every, merge=args.merge, item_rates=rates)
    if args.iters_per_launch > 1:
        cu.lib().cu2rec_hogwild_iters_per_launch(args.iters_per_launch)
    mode = {"hogwild": cu.SGD_HOGWILD, "serial": cu.SGD_SERIAL, "ordered": cu.SGD_ORDERED}[args.mode]
    log("model + ratings resident in HBM")
    rmse0 = job.loss(d_test)["rmse"]
    log("initial test rmse %.6f" % rmse0)

    it = 0
    resident_fault = None
    try:
        job.run(hyper, 42, it, args.warmup, mode)
        torch.cuda.synchronize()
        # a resident launch that could not get the GPU to itself reports at the next call into the library: ask now
        # (not with a loss pass: that would leave the caches in another state for the first timed launch)
        from cu2rec_amd._lib import check
        check(cu.lib().cu2rec_check_faults())
    except cu.Cu2recError as e:
        if "resident" not in str(e) or world > 1:
            raise
        # never silently: say so, and measure the one-launch-per-iteration kernel instead of nothing at all
        resident_fault = str(e)
        log("RESIDENT LAUNCH FAULT (%s) -- falling back to CU2REC_RESIDENT=0 for this run" % resident_fault)
        cu.lib().cu2rec_hogwild_resident(0)
        eng = Engine(train.rows, train.cols, f, train.global_bias, device=device)
        job = ShardedSGD(eng, d_train, user_offset=user_offset, sync_every=sync_every, merge=args.merge, item_rates=rates)
        job.run(hyper, 42, it, args.warmup, mode)
        torch.cuda.synchronize()
    it += args.warmup
    log("warmup done")
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    job.run(hyper, 42, it, args.steps, mode)
    job.finish_pending()  # an overlapped item-factor all-reduce still in flight belongs to the timed work
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    it += args.steps
    log("timed region: %d steps in %.4f s" % (args.steps, elapsed))
    final = job.loss(d_test)  # the model after warmup + steps iterations, before the untimed extra passes below
    final_iterations = it
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- SGD kernel duration: HIP events around single launches on the launch stream (untimed extra pass).
    # Hogwild calls of `sync_every` iterations are ONE resident launch when the user rows fit the register file
    # (cu2rec_amd/csrc/resident.hip); otherwise a launch is one iteration (or --iters-per-launch of them).
    import ctypes
    blocks_c, upg_c = ctypes.c_int(0), ctypes.c_int(0)
    resident = (args.mode == "hogwild" and args.iters_per_launch == 1 and
                cu.lib().cu2rec_hogwild_resident_plan(train.rows, f, sync_every, ctypes.byref(blocks_c), ctypes.byref(upg_c)) == 1)
    kernel_name = "sgd_resident_kernel" if resident else "sgd_%s_kernel" % args.mode
    k_launch = sync_every if resident else (max(args.iters_per_launch, 1) if args.mode == "hogwild" else 1)
    n_s = max(args.kernel_samples // (8 if resident else 1), 1)

    def time_launches(n_samples, iters_per_launch, start_it):
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_samples)]
        for a, b in evs:
            a.record()
            eng.sgd(d_train, hyper, 42, start_it, iters_per_launch, mode, True, user_offset)
            b.record()
            start_it += iters_per_launch
        torch.cuda.synchronize()
        ms = [a.elapsed_time(b) for a, b in evs]
        return float(np.mean(ms)), float(np.min(ms)), start_it

    kernel_ms, kernel_ms_min, it = time_launches(n_s, k_launch, it)
    log("%s: %d iteration(s) per launch, avg %.2f us (min %.2f us) per launch" % (kernel_name, k_launch, 1e3 * kernel_ms,
                                                                                   1e3 * kernel_ms_min))
    # ---- the streaming form beside it (one launch per iteration, user rows through HBM): same data, same model
    streaming = None
    if resident:
        prev = cu.lib().cu2rec_hogwild_resident(0)
        try:
            s_ms, s_min, it = time_launches(max(args.kernel_samples, 1), 1, it)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            eng.sgd(d_train, hyper, 42, it, 500, mode, True, user_offset)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t1
            it += 500
        finally:
            cu.lib().cu2rec_hogwild_resident(prev)
        s_bytes = users_active * (16 * f + 32)
        streaming = {"mode": "hogwild, one launch per iteration (CU2REC_RESIDENT=0): user rows stream through HBM",
                     "kernel": "sgd_hogwild_kernel", "value": users_active * 500 / dt, "unit": "updates/s",
                     "ms_per_step": 1e3 * dt / 500, "steps": 500, "kernel_avg_us": 1e3 * s_ms, "kernel_min_us": 1e3 * s_min,
                     "algorithmic_bytes_per_launch": s_bytes, "achieved_GBs": s_bytes / (s_ms * 1e-3) / 1e9,
                     "frac_of_hbm_peak": s_bytes / (s_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
        log("streaming kernel: avg %.2f us per launch, %.4f ms/step" % (1e3 * s_ms, streaming["ms_per_step"]))
    # ---- the fused loss pass (train set), timed the same way: the other kernel of the path
    le = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
    for a, b in le:
        a.record()
        eng.loss(d_train)
        b.record()
    torch.cuda.synchronize()
    loss_ms = float(np.min([a.elapsed_time(b) for a, b in le]))
    loss_hbm = 8 * train.nnz + 4 * f * (train.rows + train.cols) + 4 * (train.rows + train.cols) + 4 * (train.rows + 1)
    loss_kernel = {"kernel": "loss_fused_kernel", "ratings": train.nnz, "ms": loss_ms,
                   "ratings_per_s": train.nnz / (loss_ms * 1e-3),
                   "bound": "L2 gather of item rows (nnz * 4f bytes) on top of the HBM stream",
                   "algorithmic_hbm_bytes": loss_hbm, "hbm_GBs": loss_hbm / (loss_ms * 1e-3) / 1e9,
                   "gather_bytes": 4 * f * train.nnz, "gather_GBs": 4 * f * train.nnz / (loss_ms * 1e-3) / 1e9,
                   "note": "includes the 64 KB partial-sum copy back and host reduction (one call of cu2rec_loss)"}

    # ---- the exact mode beside it: same data, fresh model, sequential semantics (bit-identical to the CPU oracle)
    ordered = None
    if world == 1 and args.ordered_steps > 0 and args.mode != "ordered":
        eng_o = Engine(train.rows, train.cols, f, train.global_bias, device=device)
        eng_o.sgd(d_train, hyper, 42, 0, 64, cu.SGD_ORDERED)  # warm-up incl. schedule workspace creation
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        eng_o.sgd(d_train, hyper, 42, 64, args.ordered_steps, cu.SGD_ORDERED)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        ordered = {"mode": "ordered (sequential semantics, deterministic; bit-identical to the CPU oracle in tests)",
                   "value": users_active * args.ordered_steps / dt, "unit": "updates/s",
                   "ms_per_step": 1e3 * dt / args.ordered_steps, "steps": args.ordered_steps,
                   "test_rmse": eng_o.loss(d_test)["rmse"], "iterations_run": 64 + args.ordered_steps}
        log("ordered mode: %.3f ms/step" % ordered["ms_per_step"])
        del eng_o
    # ---- opt-in Hogwild launch blocking beside it (4 updates per user per launch, user row in registers)
    blocked = None
    if world == 1 and args.mode == "hogwild" and args.iters_per_launch == 1 and args.ordered_steps > 0:
        eng_b = Engine(train.rows, train.cols, f, train.global_bias, device=device)
        cu.lib().cu2rec_hogwild_iters_per_launch(4)
        try:
            eng_b.sgd(d_train, hyper, 42, 0, 100, cu.SGD_HOGWILD)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            eng_b.sgd(d_train, hyper, 42, 100, 2000, cu.SGD_HOGWILD)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t1
        finally:
            cu.lib().cu2rec_hogwild_iters_per_launch(1)
        blocked = {"mode": "hogwild, 4 iterations per launch (opt-in: users up to 3 iterations apart inside a launch)",
                   "value": users_active * 2000 / dt, "unit": "updates/s", "ms_per_step": 1e3 * dt / 2000, "steps": 2000,
                   "test_rmse": eng_b.loss(d_test)["rmse"], "iterations_run": 2100}
        log("hogwild x4 per launch: %.4f ms/step" % blocked["ms_per_step"])
        del eng_b
    bytes_per_update = 16 * f + 32
    alg_bytes = users_active * bytes_per_update * k_launch
    achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
    total_users = users_active * world
    if world > 1:
        tu = torch.tensor([float(users_active)], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(tu)
        total_users = float(tu.item())
    value = total_users * args.steps / elapsed

    if rank == 0:
        line = {
            "metric": "ratings/sec (SGD updates/sec)", "value": value, "unit": "updates/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": args.scaling if world > 1 else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s-shape synthetic ratings (users %d, items %d, train nnz %d, test nnz %d per GPU), "
                                   "f=%d, lr .01, reg .02, mode %s%s" % (args.workload, train.rows, train.cols, train.nnz, test.nnz, f,
                                                                           args.mode, "" if args.iters_per_launch == 1 else
                                                                           " x%d iterations per launch" % args.iters_per_launch),
                       "updates_per_step_per_gpu": users_active, "sync_every": sync_every if world > 1 else None,
                       "merge": args.merge if world > 1 else None, "exchanges": job.exchanges},
            "test_rmse": final["rmse"], "test_rmse_initial": rmse0, "iterations_run": final_iterations,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel": kernel_name, "kernel_avg_us": 1e3 * kernel_ms,
                         "kernel_min_us": 1e3 * kernel_ms_min, "algorithmic_bytes_per_launch": alg_bytes,
                         "bytes_per_update": bytes_per_update, "iterations_per_launch": k_launch},
        }
        if resident_fault:
            line["resident_fault"] = resident_fault
        if resident:
            line["roofline"]["note"] = (
                "one persistent launch = %d iterations; %d workgroups x 32 groups keep %d user rows each in registers, so "
                "the 8f bytes per update of user-row traffic in the algorithmic count never reach HBM: frac > 1 means the "
                "kernel beats the roofline of the streaming formulation, `traffic` is what HBM really moved"
                % (k_launch, blocks_c.value, upg_c.value))
        traffic = measured_traffic(args.workload, f, kernel_name)
        if traffic:
            line["roofline"]["traffic"] = traffic["bytes_per_launch"]
            line["roofline"]["traffic_source"] = traffic["source"]
        line["loss_kernel"] = loss_kernel
        if streaming:
            line["hogwild_streaming_mode"] = streaming
        if ordered:
            line["ordered_mode"] = ordered
        if blocked:
            line["hogwild_blocked_mode"] = blocked
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(train, test, f, hyper)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
