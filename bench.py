#!/usr/bin/env python3
"""Benchmark of the cu2rec hot path on MI355X: SGD updates/sec (+ test RMSE), ML-20M shape, f=100.

  python bench.py --gpus N --steps K --warmup W          (N > 1 with no launcher: starts its N ranks itself, see main())
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one reference iteration: one SGD update for every user with at least one rating (sgd.cu:27-37).
`value` is measured in the mode that meets the north-star tolerance -- `blocksolve`: mf_sequential.cu's semantics with
the hot item chains solved block-wise (cu2rec_amd/csrc/blocksolve.hip), within 1e-4 test RMSE of the sequential result
(`rmse_gap_vs_sequential` in the line, taken at equal iterations against the exact ordered mode) -- and the racy
Hogwild modes (the reference GPU kernel's own semantics: resident launches, streaming launches) are reported beside it
with THEIR gap.  Inputs (CSR, P, Q, biases) are resident in HBM before the timed region.  The timed region is exactly
K steps between barrier + device synchronisation; a short region (< 0.25 s) is repeated and the MEAN region reported (all
timed steps over all timed seconds; the median region beside it).

N > 1: one process per GPU, ONE dataset whose users are sharded across the ranks (strong scaling, BASELINE.json
configs[3]; --scaling weak gives every rank its own full-size population), trained by the C++ driver
(cu2rec_amd/csrc/sharded.cpp): the replicas of Q / item_bias are reconciled by one ncclAllReduce of the unpadded item
deltas every --sync-every steps (default one epoch = nnz / users), inside the timed region.

Prints ONE JSON line (rank 0).  Extra objects:
  roofline     for the timed mode: algorithmic bytes (16 f + 32 per update) / duration measured
               here with HIP events on the launch stream; frac <= 1 by construction.  Per-kernel durations: profiles/.
  cpu_baseline the reference's own CPU twin (oracle/_ref/mf_cpu, kind "reference") timed on a bounded sample of the
               same workload on this box's host cores, plus the oracle port with the counter-based sampler ("port").
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

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
GATHER_PEAK_GBS = 8600.0     # same guide, "Indexed rows": random rows of a table that lives in the Infinity Cache
MIN_TIMED_S = 0.25           # every reported rate rests on at least this much timed work (regions repeated: the mean; side-mode calls: the median)


def load_dataset(name, seed, rank, barrier):
    """Synthetic set with the named shape; generated once per box and cached under /tmp."""
    from cu2rec_amd import synth
    from cu2rec_amd.api import HostCSR
    path = os.path.join(tempfile.gettempdir(), "cu2rec_synth_%s_%d.npz" % (name, seed))
    if rank == 0 and not os.path.exists(path):
        tr, te = synth.make_named(name, seed=seed)
        import threading
        tmp = path + ".%d.%d.tmp.npz" % (os.getpid(), threading.get_ident())  # (several processes / threads may find the cache empty at once)
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


def host_description():
    """CPU model of this box and the compilers behind the two CPU baselines."""
    model, cores = "unknown", os.cpu_count()
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass

    def version(cmd):
        try:
            return subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=20).stdout.strip().splitlines()[0]
        except Exception:  # (no compiler on the box: the binaries travelled prebuilt)
            return "unavailable on this box (binary prebuilt in the build container)"
    return {"cpu_model": model, "logical_cpus": cores,
            "port_compiler": version(["gcc", "--version"]) + "; flags -std=c99 -O3 -ffp-contract=off -fno-fast-math -fopenmp (oracle/Makefile)",
            "reference_compiler": version(["/opt/rocm/bin/hipcc", "--version"]) + "; hipify-perl + hipcc -std=c++14 host compile of the reference's unity "
                                  "source (oracle/build_ref.py; the image has no nvcc)"}


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
    out["host"] = host_description()
    return out


# A user-sharded run is NOT the sequential run (the item side is reconciled once per period): what its test RMSE differs by from the
# N = 1 (= mf_sequential.cu) result, per N -- measured with the PRODUCT's driver at full size (N ranks as threads of one process on one
# GPU) and pinned by tests/test_gpu_sharded.py (profiles/r06_sharded_equal_schedule.txt).
SHARDED_TOLERANCE = {
    "against": "test RMSE of the N = 1 run (= the sequential result), block-solve per shard, `adaptive` merge, one exchange per epoch",
    "fixed iterations at lr .01": {
        "ml-20m f=100, 1,000 iterations": {"2": {"measured": -1e-5, "accepted": 1e-4}, "4": {"measured": 1.7e-4, "accepted": 3e-4},
                                           "8": {"measured": 3.9e-4, "accepted": 6e-4}},
        "netflix f=128, 660 iterations": {"8": {"measured": -1.2e-3, "accepted": 1.6e-3}}},
    "converged, 8,000 iterations: [end point under the run's OWN patience decisions, end point under the N = 1 run's learning-rate history, best checkpoint]": {
        "ml-20m f=100": {"2": [-4.3e-3, -4.3e-3, -4.3e-5], "4": [-3.5e-3, -8.0e-3, -2.6e-4], "8": [-7.7e-3, -1.06e-2, -2.7e-4]},
        "netflix f=128": {"2": [-2.8e-3, -2.6e-3, 2.2e-4], "4": [-9.5e-4, -6.1e-3, 1.4e-3], "8": [-5.0e-3, -8.9e-3, 1.8e-3]},
        "accepted (tests)": {"ml-20m f=100 N=8": 1.3e-2, "netflix f=128 N=8": 1.1e-2}},
    "cause": "negative = the sharded run's test RMSE is LOWER: the gap opens where N = 1 overfits (ml-20m: 0.8098 at 1,000 iterations -> 0.8205 frozen); "
             "hot item rows merged as a weighted mean of N shard-local results overfit more slowly.  Not the patience rule (equal LR histories "
             "widen the gap), not the exchange period (2 ... 115 iterations: same gap), not the adaptive constant (2 ... 20); an all-reduce "
             "every iteration with the constant scaled to the period (the library's rule since round 6) leaves -8.5e-4 / -1.35e-3 / -1.5e-3 at N = 2 / 4 / 8",
    "north_star_bar": 1e-4, "meets_north_star_bar": "N <= 2 only (fixed iterations)",
    "survey_8e_bar": "converged <= 1e-3 of N = 1: NOT met at the end point under either schedule; met at the best checkpoint on the ML-20M shape only"}

# rocprofv3 --pmc summaries (tools/pmc_summary.py) of THIS workload and mode, if one is committed: (workload, factors, mode) -> file
def _latest(pattern):
    """The newest round's committed file of that name (profiles/rNN_...), or None."""
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    return os.path.relpath(found[-1], ROOT) if found else None


PMC_PROFILES = {("ml-20m", 100, "blocksolve"): _latest("r[0-9][0-9]_pmc_blocksolve_ml20m_f100.json"),
                ("netflix", 128, "blocksolve"): _latest("r[0-9][0-9]_pmc_blocksolve_netflix_f128.json")}
KERNEL_STATS = {("ml-20m", 100, "blocksolve"): _latest("r[0-9][0-9]_kernel_stats_blocksolve_ml20m_f100.csv"),
                ("netflix", 128, "blocksolve"): _latest("r[0-9][0-9]_kernel_stats_blocksolve_netflix_f128.csv")}
LOSS_PMC = {("ml-20m", 100): _latest("r[0-9][0-9]_pmc_loss_train_ml20m_f100.json")}
L2_GATHER_TBS = 16.8  # MI355X_MICROARCH.md, "Indexed rows": rows shared by every workgroup (the XCD's L2), chip-wide lower bound
# the kernels of one SGD iteration of a mode (the schedule kernels run once per batch of 64 iterations: counted per iteration below)
ITERATION_KERNELS = {"blocksolve": ("bs_gram_kernel", "bs_solve_kernel", "bs_update_kernel", "sgd_ordered_kernel")}


def profile_provenance(files):
    """Do the committed profile files belong to THIS build?  profiles/rNN_profile_meta.json (tools/source_digest.py --write) records the
    kernel sources' digest the round's summaries were taken from; a summary of another build beside a live timing is a mixed roofline."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from source_digest import kernel_source_digest
    now, meta_file = kernel_source_digest(), _latest("r[0-9][0-9]_profile_meta.json")
    out = {"build_source_digest": now, "profile_meta": meta_file, "profile_source_digest": None, "profiles_match_build": None}
    if meta_file:
        with open(os.path.join(ROOT, meta_file)) as fh:
            meta = json.load(fh)
        covered = all(os.path.basename(f) in meta.get("covers", []) for f in files if f)
        out["profile_source_digest"] = meta.get("source_digest") if covered else None
        out["profiles_match_build"] = bool(covered and meta.get("source_digest") == now)
    return out


def profile_traffic(workload, factors, mode):
    """HBM-side bytes per iteration from the committed PMC summary of this workload (FETCH_SIZE doubled per
    MI355X_MICROARCH.md; separate --pmc passes), or (None, why): the sum over the iteration's kernels of their corrected
    bytes per launch.  Never another workload's counters."""
    rel = PMC_PROFILES.get((workload, factors, mode))
    if rel is None:
        return None, "no rocprofv3 --pmc summary committed for workload %s f=%d mode %s" % (workload, factors, mode)
    path = os.path.join(ROOT, rel)
    if not os.path.exists(path):
        return None, "%s not found" % rel
    with open(path) as fh:
        prof = json.load(fh)
    total, parts = 0.0, []
    for name in ITERATION_KERNELS.get(mode, ()):
        for kernel, rec in prof.items():
            if name + "<" in kernel or name + "(" in kernel:
                b = rec.get("hbm_bytes_per_launch_corrected")
                if b is not None:
                    total += b
                    parts.append("%s %.1f MB" % (name, b / 1e6))
    if not parts:
        return None, "%s lists none of the iteration's kernels" % rel
    return total, ("%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, KiB units, FETCH doubled; EVENT TOPOLOGY: under a counter pass the "
                   "library forks / joins the side stream with events, the timed region uses the device-side gate and join -- same kernels, same bytes; "
                   "per launch: %s)" % (rel, ", ".join(parts)))


def profile_kernels(workload, factors, mode):
    """Per kernel of one iteration, from the COMMITTED rocprofv3 summaries of this workload: average duration (kernel stats), HBM-side
    bytes per launch (PMC passes, corrected as above), bytes / duration against the HBM peak.  None without both files."""
    import csv
    stats, pmc = KERNEL_STATS.get((workload, factors, mode)), PMC_PROFILES.get((workload, factors, mode))
    if not stats or not pmc or not os.path.exists(os.path.join(ROOT, stats)) or not os.path.exists(os.path.join(ROOT, pmc)):
        return None
    with open(os.path.join(ROOT, pmc)) as fh:
        counters = json.load(fh)
    out = []
    with open(os.path.join(ROOT, stats)) as fh:
        for row in csv.DictReader(fh):
            for name in ITERATION_KERNELS.get(mode, ()):
                if name + "<" in row["Name"] or name + "(" in row["Name"]:
                    us = float(row["AverageNs"]) / 1e3
                    b = next((rec.get("hbm_bytes_per_launch_corrected") for k, rec in counters.items() if name + "<" in k or name + "(" in k), None)
                    hit = next((rec.get("l2_hit_rate") for k, rec in counters.items() if name + "<" in k or name + "(" in k), None)
                    out.append({"kernel": name, "avg_us": us, "calls": int(row["Calls"]), "hbm_bytes_per_launch": b, "l2_hit_rate": hit,
                                "GBs": (b / (us * 1e-6) / 1e9) if b else None, "frac_of_hbm_peak": (b / (us * 1e-6) / 1e9 / HBM_PEAK_GBS) if b else None})
    return {"from": [stats, pmc], "kernels": out} if out else None


def log(msg):
    sys.stderr.write("[bench %7.1fs] %s\n" % (time.perf_counter() - T_START, msg))
    sys.stderr.flush()


T_START = time.perf_counter()


class HipEvents:
    """HIP events on the stream the library launches on (the null stream): hipEventRecord / hipEventElapsedTime."""

    def __init__(self):
        import ctypes as C
        self.C = C
        self.hip = C.CDLL("libamdhip64.so")
        self.hip.hipEventCreate.argtypes = [C.POINTER(C.c_void_p)]
        self.hip.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]
        self.hip.hipEventSynchronize.argtypes = [C.c_void_p]
        self.hip.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), C.c_void_p, C.c_void_p]

    def new(self):
        e = self.C.c_void_p()
        assert self.hip.hipEventCreate(self.C.byref(e)) == 0
        return e

    def record(self, e):
        assert self.hip.hipEventRecord(e, None) == 0

    def ms(self, a, b):
        assert self.hip.hipEventSynchronize(b) == 0
        out = self.C.c_float()
        assert self.hip.hipEventElapsedTime(self.C.byref(out), a, b) == 0
        return float(out.value)


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
    ap.add_argument("--mode", default="blocksolve", choices=["blocksolve", "hogwild", "serial", "ordered"])
    ap.add_argument("--sync-every", type=int, default=0, help="steps between item-factor all-reduces (0 = one epoch)")
    ap.add_argument("--merge", default="adaptive", choices=["mean", "sum", "weighted", "adaptive"])
    ap.add_argument("--scaling", default="strong", choices=["weak", "strong"],
                    help="strong (default): ONE dataset, users sharded across the ranks (BASELINE.json configs[3]); weak: "
                         "every rank holds its own full-size user population")
    ap.add_argument("--seed", type=int, default=20240917)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-side-modes", action="store_true", help="skip the untimed legs (Hogwild, ordered, loss kernel)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` with no launcher: start the N ranks as FRESH child processes (torch.distributed.run, one
        # rank per GPU) before this process has imported torch or touched HIP -- nothing that initialised a GPU is ever
        # re-executed -- relay their output (rank 0 prints the JSON line) and leave with their exit code.
        import socket
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        log("starting %d ranks: %s" % (args.gpus, " ".join(cmd)))
        sys.exit(subprocess.run(cmd).returncode)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        args.gpus = world  # (started by a launcher with another world size: the launcher is right)
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the cu2rec_amd hot path has no CPU fallback")
    # CU2REC_BENCH_BACKEND=gloo lets several ranks share one GPU (plumbing check on a 1-GPU box; the item exchange then
    # goes through the host); the real multi-GPU run is RCCL over xGMI, one rank per GPU
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
    from cu2rec_amd import sharded
    from cu2rec_amd._lib import check
    L = cu.lib()
    check(L.cu2rec_set_device(dev_index))
    log("torch + library loaded, device %s" % torch.cuda.get_device_name(device))
    if world > 1 and backend != "nccl":
        L.cu2rec_hogwild_resident(0)  # ranks share a GPU here: a resident launch needs the GPU to itself
    full_train, full_test = load_dataset(args.workload, args.seed, rank, barrier)
    log("dataset ready: %d users, %d items, %d train ratings" % (full_train.rows, full_train.cols, full_train.nnz))
    f = args.factors
    hyper = (0.01, 0.02, 0.02, 0.02, 0.02)  # preprocessing/create_config.py:25-32
    train, test, user_offset = full_train, full_test, 0
    if world > 1:
        if args.scaling == "strong":
            user_offset, _, train, test = sharded.shard_of(full_train, full_test, rank, world)
        else:
            user_offset = rank * full_train.rows  # rank r's users are users [r * rows, (r + 1) * rows) of the population
    users_active = int(np.count_nonzero(np.diff(train.indptr)))

    # the communicator of the data path: RCCL created by the library itself (rank 0's id broadcast through torch), or the
    # gloo debugging aid as a callback
    if world > 1 and backend != "nccl":
        import ctypes as C
        hip = C.CDLL("libamdhip64.so")
        hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]

        def host_allreduce(ctx, buf, count, is_double, stream):
            host = np.empty(count, np.float64 if is_double else np.float32)
            if hip.hipDeviceSynchronize() != 0 or hip.hipMemcpy(host.ctypes.data, buf, host.nbytes, 2) != 0:
                return 1
            dist.all_reduce(torch.from_numpy(host))
            return 0 if hip.hipMemcpy(buf, host.ctypes.data, host.nbytes, 1) == 0 else 1
        comm = sharded.Comm(rank, world, allreduce=host_allreduce)
    else:
        comm = sharded.Comm(rank, world, share=sharded.share_through_torch(device) if world > 1 else None)

    def fresh_model():
        if world > 1 and args.scaling == "strong":  # every rank keeps its slice of the reference's seed-42 initialisation
            P0 = cu.api.initialize_normal_array(full_train.rows * f, f).reshape(full_train.rows, f)
            ub0 = cu.api.initialize_normal_array(full_train.rows, f)
            u0, u1 = user_offset, user_offset + train.rows
            return cu.Model(train.rows, train.cols, f, train.global_bias, P=P0[u0:u1], user_bias=ub0[u0:u1])
        return cu.Model(train.rows, train.cols, f, train.global_bias)

    model = fresh_model()
    d_train, d_test = cu.DeviceCSR(train), cu.DeviceCSR(test)
    # N = 1: nothing to exchange -- calls of 500 iterations, the stretch between two loss checks of the reference's loop
    # (check_error = 500, config.h:41-51, training.cu:118) and what cu2rec_train issues
    job = sharded.ShardJob(comm, model, d_train, user_offset=user_offset, sync_every=args.sync_every if world > 1 else 500,
                           merge=args.merge)
    info = job.info()
    mode = args.mode
    log("model + ratings resident in HBM; mode %s, sync every %d" % (mode, info["sync_every"]))
    rmse0 = job.loss(d_test)["rmse"]
    log("initial test rmse %.6f" % rmse0)

    def reduce_max(values):
        """Element-wise MAX over the ranks of a list of floats (outside every clock)."""
        if world == 1:
            return [float(v) for v in values]
        t = torch.tensor(list(values), dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return [float(v) for v in t.cpu().tolist()]

    def warm_up(j, m):
        """W untimed steps.  With several ranks the warm-up also carries TWO full-size exchanges (the wire buffer is always the whole
        item side: RCCL sets up its channels for that message size on the first, the second runs in the steady state) and leaves
        the cadence at zero, so that no timed region holds the communicator's first large collective."""
        if world > 1 and args.warmup >= 2:
            j.run(hyper, 42, 0, args.warmup - 1, m)
            j.exchange()
            j.run(hyper, 42, args.warmup - 1, 1, m)
            j.exchange()
        else:
            j.run(hyper, 42, 0, args.warmup, m)
            if world > 1:
                j.exchange()
        torch.cuda.synchronize()
        check(L.cu2rec_check_faults())

    it = 0
    warm_up(job, mode)
    it += args.warmup
    warm_exchanges = job.info()["exchanges"]
    log("warmup done (%d exchange(s) inside it)" % warm_exchanges)
    ev = HipEvents()
    e0, e1 = ev.new(), ev.new()

    def timed_region(j, m, start_it):
        """EXACTLY K steps between barrier + device synchronisation on both sides.  The wall clock stops after THIS rank's
        synchronisation -- the trailing barrier is outside it -- and the MAX over ranks of every region is taken afterwards,
        outside every clock (reduce_max)."""
        barrier()
        torch.cuda.synchronize()
        x0 = j.info()["exchanges"]
        t0 = time.perf_counter()
        ev.record(e0)
        j.run(hyper, 42, start_it, args.steps, m)
        ev.record(e1)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        barrier()
        return dt, 1e-3 * ev.ms(e0, e1), j.info()["exchanges"] - x0

    regions = [timed_region(job, mode, it)]
    it += args.steps
    # A short region is mostly launch + synchronisation noise: every region is EXACTLY K steps, and regions are repeated until
    # a quarter of a second has been timed in all (at least 5 when one region is shorter than that, at most 400).  Reported: ALL
    # timed steps over ALL timed seconds, i.e. the MEAN region -- calls of K steps run out of schedule windows of 64 iterations
    # (ordered.hip), so one region in three carries the next window's schedule kernels and the others none, and with several ranks
    # one region in sync_every / K carries an exchange: the median region (on the line as timed_region_s_median) would pick the
    # cheap ones.
    first_all = reduce_max([regions[0][0]])[0]  # every rank must take the same decision: the regions are bracketed by barriers
    n_regions = 1 if first_all >= MIN_TIMED_S else int(min(400, max(5, np.ceil(MIN_TIMED_S / max(first_all, 1e-6)))))
    if world > 1:  # (whole exchange periods: the mean region then carries the exchanges' true share)
        per_period = int(np.ceil(info["sync_every"] / max(args.steps, 1)))
        n_regions = int(min(400, max(n_regions, per_period) if per_period <= 400 else n_regions))
    for _ in range(n_regions - 1):
        regions.append(timed_region(job, mode, it))
        it += args.steps
    check(L.cu2rec_check_faults())
    wall_all = reduce_max([r[0] for r in regions])  # per region: the slowest rank's wall time
    elapsed = float(np.mean(wall_all))
    elapsed_dev = float(np.mean([r[1] for r in regions]))
    elapsed_median = float(np.median(wall_all))
    regions_with_exchange = int(sum(1 for r in regions if r[2] > 0))
    log("timed: %d region(s) of %d steps, mean %.6f s (device %.6f s), median %.6f s, %d with an exchange"
        % (len(regions), args.steps, elapsed, elapsed_dev, elapsed_median, regions_with_exchange))
    x_stats = job.exchange_stats()  # (every stream is synchronised here: every exchange's event pair has completed)
    per_rank = None
    if world > 1:
        mine = {"rank": rank, "device_seconds_mean_region": elapsed_dev, "wall_seconds_mean_region": float(np.mean([r[0] for r in regions])),
                "users": users_active, "exchange_seconds_mean": x_stats["mean_seconds"], "comm": comm.info()}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
    job.exchange()
    final = job.loss(d_test)
    final_iterations = it
    exchanges = job.info()["exchanges"]
    total_users = float(job.info()["users_total"])
    value = total_users * args.steps / elapsed
    bytes_per_update = 16 * f + 32
    alg_bytes = users_active * bytes_per_update * args.steps
    achieved = alg_bytes / elapsed_dev / 1e9

    side = {}
    if world > 1 and mode == "blocksolve" and not args.no_side_modes:
        # ---- the same shards in the throughput mode (Hogwild per shard), same exchange cadence and merge: the 1 -> N curve of the
        # driver's scaling run then shows both the certified mode and the racy one.  Every rank runs the same number of
        # barrier-bracketed regions (n_regions was agreed on above), so the ranks stay in lockstep.
        hog_model = fresh_model()
        hog_job = sharded.ShardJob(comm, hog_model, d_train, user_offset=user_offset, sync_every=args.sync_every, merge=args.merge)
        warm_up(hog_job, "hogwild")
        hit, hregs = args.warmup, []
        for _ in range(n_regions):
            hregs.append(timed_region(hog_job, "hogwild", hit)[0])
            hit += args.steps
        check(L.cu2rec_check_faults())
        h_el = float(np.mean(reduce_max(hregs)))  # (the mean region, like the headline: the exchanges' share is in it)
        hog_job.exchange()
        h_final = hog_job.loss(d_test)
        side["hogwild_sharded_mode"] = {
            "mode": "hogwild per shard (sgd.cu's racy semantics), same shards, exchange cadence and merge as the headline",
            "value": float(hog_job.info()["users_total"]) * args.steps / h_el, "unit": "updates/s", "ms_per_step": 1e3 * h_el / args.steps,
            "timed_regions": len(hregs), "test_rmse": h_final["rmse"], "iterations_run": hit,
            "note": "not a parity claim: Hogwild is 1e-3 away from mf_sequential.cu while the model moves (DESIGN.md section 5)"}
        hog_job.close()
        del hog_model
    if world == 1 and not args.no_side_modes:
        # ---- the sequential result at equal iterations: the exact ordered mode (bit-identical to the CPU oracle in tests)
        def run_fresh(m, iters, policy=None, call=500):
            """A fresh model in mode m: to `iters` iterations for its test RMSE at the headline's iteration count, in calls of the
            cadence the mode is defined for (`call` = 500 iterations: the reference's stretch between two loss checks, config.h
            check_error, and what cu2rec_train issues -- for the resident Hogwild form that is ONE launch); every full call is timed
            by HIP events, and full calls go on (past `iters`, after the model's state has been read) until a quarter of a second
            of device time has been timed.  Reported: the median full call, per step."""
            mod = fresh_model()
            prev = L.cu2rec_hogwild_resident(policy) if policy is not None else None
            per_step_wall, per_step_dev, timed_dev = [], [], 0.0

            def one_call(start, n):
                a, b = ev.new(), ev.new()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                ev.record(a)
                mod.sgd(d_train, hyper, 42, start, n, mode=m)
                ev.record(b)
                torch.cuda.synchronize()
                return time.perf_counter() - t1, 1e-3 * ev.ms(a, b)
            try:
                mod.sgd(d_train, hyper, 42, 0, 64, mode=m)  # warm-up incl. workspace creation
                done = 64
                while done < iters:
                    n = min(call, iters - done)
                    dt, dev = one_call(done, n)
                    if n == call:
                        per_step_wall.append(dt / n), per_step_dev.append(dev / n)
                        timed_dev += dev
                    done += n
                rmse = mod.loss(d_test)["rmse"]
                while timed_dev < MIN_TIMED_S or len(per_step_dev) < 3:
                    dt, dev = one_call(done, call)
                    per_step_wall.append(dt / call), per_step_dev.append(dev / call)
                    timed_dev += dev
                    done += call
            finally:
                if prev is not None:
                    L.cu2rec_hogwild_resident(prev)
            check(L.cu2rec_check_faults())
            del mod
            wall, dev = float(np.median(per_step_wall)), float(np.median(per_step_dev))
            return rmse, users_active / wall, 1e3 * wall, 1e3 * dev, len(per_step_dev)

        gap_iters = final_iterations
        seq_rmse, seq_rate, seq_ms, _, seq_calls = run_fresh("ordered", gap_iters)
        side["rmse_gap_vs_sequential"] = {
            "iterations": gap_iters, "rmse": final["rmse"], "sequential_rmse": seq_rmse, "gap": abs(final["rmse"] - seq_rmse),
            "tolerance": 1e-4, "sequential": "CU2REC_SGD_ORDERED: mf_sequential.cu:102-143's result (bit-identical to the CPU "
                                             "oracle, tests/test_gpu_parity.py), same data, same sample stream, same iterations"}
        side["ordered_mode"] = {"mode": "ordered (sequential semantics, exact)", "value": seq_rate, "unit": "updates/s",
                                "ms_per_step": seq_ms, "test_rmse": seq_rmse, "iterations_run": gap_iters, "timed_calls_of_500": seq_calls}
        log("ordered mode: %.3f ms/step, gap of the timed mode %.2e" % (seq_ms, side["rmse_gap_vs_sequential"]["gap"]))
        if mode == "blocksolve":
            # ---- the racy modes beside it, with THEIR gap at the same iteration count
            for name, policy, kernel in (("hogwild_resident_mode", 2, "sgd_resident_kernel"), ("hogwild_streaming_mode", 0, "sgd_hogwild_kernel")):
                planned = L.cu2rec_hogwild_resident_plan(train.rows, f, 500, None, None) == 1
                if policy == 2 and not planned:
                    continue
                r, rate, ms, dev_ms, n_calls = run_fresh("hogwild", gap_iters, policy)
                if policy == 2:  # user rows stay in registers: what must move per update is the item side only
                    b_upd, peak, level = 8 * f + 16, GATHER_PEAK_GBS, "Infinity-Cache / fabric gather rate of random rows (MI355X_MICROARCH.md, Indexed rows)"
                else:
                    b_upd, peak, level = 16 * f + 32, HBM_PEAK_GBS, "HBM"
                ach = users_active * b_upd / (dev_ms * 1e-3) / 1e9
                side[name] = {"mode": "hogwild, %s" % ("ONE resident launch per 500 iterations (user rows in registers)" if policy == 2 else
                                                       "one launch per iteration (user rows stream through HBM)"),
                              "kernel": kernel, "value": rate, "unit": "updates/s", "ms_per_step": ms, "test_rmse": r,
                              "iterations_run": gap_iters, "timed_calls_of_500": n_calls, "rmse_gap_vs_sequential": abs(r - seq_rmse),
                              "meets_1e-4_tolerance": bool(abs(r - seq_rmse) <= 1e-4),
                              "roofline": {"bytes_per_update": b_upd, "achieved": ach, "peak": peak, "unit": "GB/s", "frac": ach / peak,
                                           "bound": level, "device_ms_per_step": dev_ms}}
                log("%s: %.4f ms/step, gap %.2e" % (name, ms, abs(r - seq_rmse)))
        if mode == "blocksolve":
            # ---- SURVEY.md section 8c Tier 2 as written: CONVERGED runs (the product's train() under the reference's schedule --
            # check every 500 iterations, patience 2, decay 0.2, training.cu:118,146-155 -- until the rate has decayed >= 3 times)
            # against the sequential result; one sampler seed live here, three per mode in the committed table
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import tier2_converged as t2
            t2_iters = 8000
            seq_rec, _ = t2.run_mode(cu, d_train, d_test, f, 42, t2_iters, "ordered")
            # the CPU oracle's own converged run in the reference's arithmetic (sequential dot, float / double loss sums): NOT recomputed
            # here (250 s of a host core) -- read from the committed record of tests/test_gpu_blocksolve.py::
            # test_converged_ml20m_blocksolve_against_the_cpu_oracle_in_the_references_own_arithmetic, which re-runs it in every GPU suite
            cpu_rec, cpu_file = None, _latest("r[0-9][0-9]_converged_vs_cpu_oracle.json")
            if cpu_file and args.workload == "ml-20m" and f == 100:
                with open(os.path.join(ROOT, cpu_file)) as fh:
                    cpu_rec = json.load(fh)
            t2_modes = {}
            for label in ("blocksolve", "hogwild-resident", "hogwild-streaming"):
                if label == "hogwild-resident" and L.cu2rec_hogwild_resident_plan(train.rows, f, 500, None, None) != 1:
                    continue
                rec, _ = t2.run_mode(cu, d_train, d_test, f, 42, t2_iters, label)
                gap = abs(rec["final_test_rmse"] - seq_rec["final_test_rmse"])
                at_cpu = next((r for i, r in rec["checks"] if cpu_rec and i == cpu_rec["cpu_oracle_f64"]["checks"][-1]["iteration"]), None)
                t2_modes[label] = {"final_test_rmse": rec["final_test_rmse"], "min_test_rmse": rec["min_test_rmse"], "decay_iterations": rec["decay_iterations"],
                                   "gap_to_cpu_oracle_at_its_iterations": None if at_cpu is None else abs(at_cpu - cpu_rec["cpu_oracle_f64"]["final_test_rmse"]),
                                   "final_learning_rate": rec["final_learning_rate"], "converged": rec["converged"], "gap_vs_sequential": gap,
                                   "within_1e-4": bool(gap <= 1e-4)}
                log("tier 2 converged, %s: gap %.2e" % (label, gap))
            check(L.cu2rec_check_faults())
            cpu_fields = {}
            if cpu_rec:
                n_cpu = cpu_rec["cpu_oracle_f64"]["checks"][-1]["iteration"]
                at = {label: next((r for i, r in rec_["checks"] if i == n_cpu), None) for label, rec_ in (("sequential", seq_rec),)}
                cpu_fields = {"cpu_oracle_file": cpu_file, "cpu_oracle_iterations": n_cpu,
                              "cpu_oracle_final_test_rmse": cpu_rec["cpu_oracle_f64"]["final_test_rmse"],
                              "cpu_oracle_final_test_rmse_float_accumulators": cpu_rec["cpu_oracle_f32"]["final_test_rmse"],
                              "cpu_oracle_decay_iterations": cpu_rec["cpu_oracle_f64"]["decay_iterations"],
                              "test_rmse_at_cpu_oracle_iterations": at["sequential"],
                              "gap_to_cpu_oracle_at_its_iterations": None if at["sequential"] is None else abs(at["sequential"] - cpu_rec["cpu_oracle_f64"]["final_test_rmse"]),
                              "cpu_oracle_arithmetic": "oracle/cu2rec_oracle.c orc_train, DOT_SEQ (mf_sequential.cu:122-125), loss sums in double (loss.cu:185-190) / "
                                                       "float (mf_sequential.cu:147-174: what the reference prints, biased low by %.1e)" % cpu_rec["float_accumulator_bias_of_the_printed_test_rmse"]}
            side["tier2_converged"] = {
                "what": "converged test RMSE against the sequential (ordered = mf_sequential.cu) run: train() under the reference's schedule, "
                        "%d iterations, sampler seed 42" % t2_iters,
                "sequential": {"final_test_rmse": seq_rec["final_test_rmse"], "min_test_rmse": seq_rec["min_test_rmse"],
                               "decay_iterations": seq_rec["decay_iterations"], "final_learning_rate": seq_rec["final_learning_rate"], "converged": seq_rec["converged"],
                               **cpu_fields},
                "modes": t2_modes, "tolerance": 1e-4,
                "headline_mode_decision": "value is timed in the fastest mode whose converged gap is <= 1e-4 on every seed: block-solve; the Hogwild forms "
                                          "miss it on every seed (three seeds per mode and the ML-1M shape: profiles/r05_tier2_converged_*.json; "
                                          "pinned by tests/test_gpu_blocksolve.py::test_tier2_converged_runs_under_the_reference_schedule_pick_the_headline_mode)"}
        # ---- the fused loss pass (train set): the other kernel of the path
        times = []
        for _ in range(5):
            a, b = ev.new(), ev.new()
            ev.record(a)
            model.loss(d_train)
            ev.record(b)
            times.append(ev.ms(a, b))
        loss_ms = float(np.min(times))
        loss_hbm = 8 * train.nnz + 4 * f * (train.rows + train.cols) + 4 * (train.rows + train.cols) + 4 * (train.rows + 1)
        # The pass's real bound: one Q row GATHERED per rating -- nnz x 4f bytes that the 4 MiB L2 of an XCD serves when
        # it holds the row and the Infinity Cache when it does not (Q is 10.7 MB: it never comes from HBM twice).  Roof = the guide's
        # measured gather rates (MI355X_MICROARCH.md, "Indexed rows": >= 16.8 TB/s from L2, 8.6 TB/s from the Infinity Cache) weighted by
        # the L2 hit rate of the committed --pmc pass of THIS workload's train-set loss.
        gather_bytes = 4 * f * train.nnz
        loss_pmc = LOSS_PMC.get((args.workload, f))
        hit = fabric = None
        if loss_pmc and os.path.exists(os.path.join(ROOT, loss_pmc)):
            with open(os.path.join(ROOT, loss_pmc)) as fh:
                rec = next((v for k, v in json.load(fh).items() if "loss_fused_kernel" in k), None)
            if rec:
                hit, fabric = rec.get("l2_hit_rate"), rec.get("hbm_bytes_per_launch_corrected")
        gather_roof = None if hit is None else 1.0 / (hit / L2_GATHER_TBS + (1.0 - hit) / (GATHER_PEAK_GBS / 1e3))
        gather_tbs = gather_bytes / (loss_ms * 1e-3) / 1e12
        side["loss_kernel"] = {"kernel": "loss_fused_kernel", "ratings": train.nnz, "ms": loss_ms, "ratings_per_s": train.nnz / (loss_ms * 1e-3),
                               "bound": "L2 / Infinity-Cache gather of one item row per rating (not HBM)",
                               "gather_bytes": gather_bytes, "gather_TBs": gather_tbs, "l2_hit_rate": hit,
                               "gather_roof_TBs": gather_roof, "frac_of_gather_roof": None if gather_roof is None else gather_tbs / gather_roof,
                               "gather_roof_from": "MI355X_MICROARCH.md 'Indexed rows': %.1f TB/s (L2, lower bound) and %.1f TB/s (Infinity Cache), weighted by the "
                                                   "L2 hit rate of %s" % (L2_GATHER_TBS, GATHER_PEAK_GBS / 1e3, loss_pmc),
                               "algorithmic_hbm_bytes": loss_hbm, "hbm_GBs": loss_hbm / (loss_ms * 1e-3) / 1e9,
                               "frac_of_hbm_peak": loss_hbm / (loss_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                               "fabric_bytes_per_launch": fabric, "fabric_over_algorithmic_hbm": None if fabric is None else fabric / loss_hbm,
                               "fabric_note": "FETCH_SIZE (doubled) + WRITE_SIZE count what leaves L2 for the fabric: item rows re-fetched past the 4 MiB L2 are "
                                              "served by the 256 MB Infinity Cache, not by HBM, so this is NOT wasted HBM traffic -- it is the miss side of the gather",
                               "note": "one call of cu2rec_loss on the train set: the fused pass, the device-side sum of its per-block partial sums and the 16-byte copy back"}

    if rank == 0:
        traffic, traffic_src = profile_traffic(args.workload, f, mode)
        kernels = {"blocksolve": "one block-solve iteration = bs_gram_kernel + bs_solve_kernel + bs_update_kernel, sgd_ordered_kernel beside them "
                                 "(per-kernel durations: profiles/rNN_kernel_stats_blocksolve_*.csv, newest round)",
                   "hogwild": "sgd_resident_kernel / sgd_hogwild_kernel", "ordered": "sgd_ordered_kernel", "serial": "sgd_serial_kernel"}
        line = {
            "metric": "ratings/sec (SGD updates/sec)", "value": value, "unit": "updates/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s-shape synthetic ratings (users %d, items %d, train nnz %d, test nnz %d%s), f=%d, lr .01, reg .02, "
                                   "mode %s" % (args.workload, full_train.rows, full_train.cols, full_train.nnz, full_test.nnz,
                                                "" if world == 1 else (", users sharded over %d ranks" % world if args.scaling == "strong"
                                                                       else " per GPU"), f, mode),
                       "mode": mode, "updates_per_step": total_users, "updates_per_step_this_rank": users_active,
                       "sync_every": info["sync_every"] if world > 1 else None, "merge": args.merge if world > 1 else None,
                       "exchanges": exchanges, "wire_bytes_per_exchange": info["wire_bytes"] if world > 1 else None,
                       "exchanges_in_warmup": warm_exchanges,
                       "sharded_run_tolerance": SHARDED_TOLERANCE if world > 1 else None},
            "timed_region_s": elapsed, "timed_region_s_mean": elapsed, "timed_region_s_median": elapsed_median, "timed_regions": len(regions),
            "timed_region_s_min_max": [round(min(wall_all), 6), round(max(wall_all), 6)],
            "timed_region_clock": "per region: barrier, device synchronisation, clock starts, K steps, device synchronisation, clock STOPS, barrier; "
                                  "per region the MAX over ranks, reduced after the last region; value = all timed steps / the sum of those maxima",
            "timed_regions_with_exchange": regions_with_exchange,
            "exchange": None if world == 1 else {
                "timed": x_stats["timed"], "mean_seconds": x_stats["mean_seconds"], "max_seconds": x_stats["max_seconds"],
                "what": "items_wire_pack -> all-reduce of wire_bytes_per_exchange -> items_wire_apply, HIP events on the exchange's stream, rank 0 "
                        "(includes the wait for the slowest rank); warm-up exchanges included in the count"},
            "rccl": comm.info(), "per_rank": per_rank,
            "test_rmse": final["rmse"], "test_rmse_initial": rmse0, "iterations_run": final_iterations,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_over_algorithmic": (traffic / (users_active * bytes_per_update)) if traffic else None,
                         "traffic_from_profile": traffic_src,
                         "launch_topology": None if mode != "blocksolve" else {
                             "timed_region": dict(zip(("form", "why"), cu.api.blocksolve_topology())),
                             "traffic_profile": "EVENT topology: a rocprofv3 --pmc pass serialises kernels across streams, so the library forks / joins the "
                                                "side stream with events there (same kernels, same bytes; the pass is for counters, never for timing)"},
                         "kernels": profile_kernels(args.workload, f, mode),
                         "profile_provenance": profile_provenance([PMC_PROFILES.get((args.workload, f, mode)), KERNEL_STATS.get((args.workload, f, mode))]),
                         "kernel": kernels.get(mode, mode), "bytes_per_update": bytes_per_update,
                         "algorithmic_bytes": alg_bytes, "device_seconds": elapsed_dev,
                         "note": "algorithmic bytes of the timed region (rank 0's updates x (16 f + 32)) / its duration by HIP events on "
                                 "the launch stream; the mode's sequential chains are latency bound, not bandwidth bound: DESIGN.md section 4"},
        }
        line.update(side)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(train, test, f, hyper)
        print(json.dumps(line), flush=True)
    job.close()
    comm.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
