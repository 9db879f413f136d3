#!/usr/bin/env python3
"""Plumbing check for an N-GPU box, to be run BEFORE the first scaling run (`bench.py --gpus N`) so that it cannot die on plumbing:

    python tools/scale_preflight.py --gpus N          (starts its N ranks itself, one per GPU, like bench.py)

  1. every rank: HIP device LOCAL_RANK, RCCL through the library's own binding -- rank 0's ncclUniqueId (cu2rec_comm_unique_id)
     broadcast through torch.distributed, ncclCommInitRank inside cu2rec_comm_create (N = 1: CU2REC_RCCL_WORLD1 forces a real one);
  2. a small user-sharded job per rank (cu2rec_shard_job: the constructor's rate / weight all-reduces in double), two periods of
     block-solve SGD with one exchange each (items_wire_pack -> ncclAllReduce over xGMI -> items_wire_apply), the global loss
     (3 doubles all-reduced);
  3. replicas bit-identical: sha256 of every rank's Q and item_bias gathered on rank 0; the global loss equal on every rank;
  4. rank 0 then runs `bin/mf -g N` on a small CSV pair (ranks forked before any HIP call, the id handed around through pipes) and
     checks the exit code and the five output files.
  5. the launcher (which never touches a GPU itself) then runs EXACTLY the driver's command -- `python bench.py --gpus 1 --steps 20
     --warmup 5`, or for N > 1 `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P
     bench.py --gpus N --steps 20 --warmup 5` -- as a fresh child in its own process group under a watchdog (--bench-timeout
     seconds: the whole group is killed and the tool exits non-zero), and checks the JSON line: n_gpus == N, steps / warmup as asked,
     what RCCL itself reports (rccl.rccl_nranks == N for N > 1), one per_rank record per rank, at least one timed region with an
     exchange.  --bench-args="..." appends flags (tests: a small workload; write it with the equals sign when the value starts with --).
Prints one PASS / FAIL line per step; exit code 0 only if all passed.  The bench line is echoed, but no number of this tool is a result."""
import argparse
import hashlib
import json
import os
import signal
import socket
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def free_port():
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


def descendants(pid):
    """Every live process below pid (children of children ...), by /proc's parent links: torch.distributed.run starts its ranks in
    sessions of their own, so a process-group kill of the launcher does not reach them."""
    parent = {}
    for name in os.listdir("/proc"):
        if name.isdigit():
            try:
                with open("/proc/%s/stat" % name) as fh:
                    fields = fh.read().rsplit(")", 1)[1].split()
                parent[int(name)] = int(fields[1])
            except (OSError, IndexError, ValueError):
                pass
    found, frontier = [], [pid]
    while frontier:
        nxt = [c for c, p in parent.items() if p in frontier and c not in found]
        found += nxt
        frontier = nxt
    return found


def run_watched(cmd, timeout_s, env=None):
    """A fresh child under a watchdog.  At the limit: the exact processes this call started -- the child and everything below it,
    listed by pid before any signal -- get SIGTERM (the launcher then takes its ranks down itself), ten seconds, then SIGKILL
    each.  -> (returncode or None, stdout)."""
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=sys.stderr, text=True, start_new_session=True, env=env)
    try:
        out, _ = proc.communicate(timeout=timeout_s)
        return proc.returncode, out
    except subprocess.TimeoutExpired:
        mine = [proc.pid] + descendants(proc.pid)
        for sig, grace in ((signal.SIGTERM, 10.0), (signal.SIGKILL, 5.0)):
            for pid in mine:
                try:
                    os.kill(pid, sig)
                except ProcessLookupError:
                    pass
            try:
                out, _ = proc.communicate(timeout=grace)
                break
            except subprocess.TimeoutExpired:
                out = ""
        else:
            proc.stdout.close()  # (someone still holds the pipe: do not wait for it)
        return None, out or ""


def driver_bench(n, timeout_s, extra):
    """Step 5: the driver's own command line for N GPUs, watched; -> (passed, detail)."""
    bench = os.path.join(ROOT, "bench.py")
    tail = ["--gpus", str(n), "--steps", "20", "--warmup", "5"] + extra
    if n == 1:
        cmd = [sys.executable, bench] + tail
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
               "--master-port", str(free_port()), bench] + tail
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "CU2REC_RCCL_WORLD1"):
        env.pop(k, None)
    rc, out = run_watched(cmd, timeout_s, env)
    if rc is None:
        return False, "watchdog: no result within %d s, process group killed: %s" % (timeout_s, " ".join(cmd))
    lines = [l for l in out.splitlines() if l.startswith("{")]
    if rc != 0 or len(lines) != 1:
        return False, "exit code %s, %d JSON line(s): %s" % (rc, len(lines), out[-1500:])
    line = json.loads(lines[0])
    print(lines[0], flush=True)
    problems = []
    if line.get("n_gpus") != n or line.get("steps") != 20 or line.get("warmup") != 5:
        problems.append("n_gpus / steps / warmup = %s / %s / %s" % (line.get("n_gpus"), line.get("steps"), line.get("warmup")))
    if not (line.get("value", 0) > 0 and "roofline" in line):
        problems.append("no value / roofline")
    if n > 1:
        rccl = line.get("rccl") or {}
        gloo = os.environ.get("CU2REC_BENCH_BACKEND", "nccl") != "nccl"  # (plumbing rehearsal on a one-GPU box: no RCCL on the data path)
        if not gloo and rccl.get("rccl_nranks") != n:
            problems.append("RCCL reports %s ranks" % rccl.get("rccl_nranks"))
        if gloo and not rccl.get("is_callback"):
            problems.append("gloo rehearsal without the callback communicator")
        ranks = line.get("per_rank") or []
        if sorted(r.get("rank") for r in ranks) != list(range(n)):
            problems.append("per_rank lists %s" % [r.get("rank") for r in ranks])
        if not gloo and any((r.get("comm") or {}).get("rccl_nranks") != n for r in ranks):
            problems.append("a rank's communicator is not %d wide" % n)
        if not line.get("timed_regions_with_exchange"):
            problems.append("no timed region held an exchange")
        if (line.get("config") or {}).get("exchanges_in_warmup", 0) < 1:
            problems.append("no exchange in the warm-up")
    return not problems, "; ".join(problems) if problems else "value %.4g %s, %d regions, %s with an exchange, exchange mean %s s" % (
        line["value"], line["unit"], line["timed_regions"], line.get("timed_regions_with_exchange"), (line.get("exchange") or {}).get("mean_seconds"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--bench-timeout", type=int, default=480)
    ap.add_argument("--bench-args", default="", help="extra flags for step 5's bench.py, space separated")
    ap.add_argument("--no-bench", action="store_true", help="steps 1-4 only")
    ap.add_argument("--bench-only", action="store_true", help="step 5 only (rehearsal of N ranks on one GPU: CU2REC_BENCH_BACKEND=gloo)")
    args = ap.parse_args()
    if "WORLD_SIZE" not in os.environ:
        if args.gpus == 1:
            os.environ["CU2REC_RCCL_WORLD1"] = "1"
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(free_port()), os.path.abspath(__file__), "--gpus", str(args.gpus)]
        rc = 0 if args.bench_only else subprocess.run(cmd).returncode  # fresh child processes: nothing that touched a GPU is re-executed
        if rc != 0 or args.no_bench:
            sys.exit(rc)
        passed, detail = driver_bench(args.gpus, args.bench_timeout, args.bench_args.split())
        print("%s  the driver's command (bench.py --gpus %d --steps 20 --warmup 5) under a %d s watchdog: %s"
              % ("PASS" if passed else "FAIL", args.gpus, args.bench_timeout, detail), flush=True)
        print("PREFLIGHT+BENCH %s" % ("OK" if passed else "FAILED"), flush=True)
        sys.exit(0 if passed else 1)

    import numpy as np
    import torch
    import torch.distributed as dist
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
    ok = True

    def report(step, passed, detail=""):
        nonlocal ok
        ok = ok and passed
        if rank == 0:
            print("%s  %s%s" % ("PASS" if passed else "FAIL", step, (": " + detail) if detail else ""), flush=True)

    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    import cu2rec_amd as cu
    from cu2rec_amd import sharded, synth
    from cu2rec_amd._lib import check
    check(cu.lib().cu2rec_set_device(local))
    try:
        comm = sharded.Comm(rank, world, share=sharded.share_through_torch(device))
        ci = comm.info()
        report("ncclCommInitRank through cu2rec_comm_create, %d rank(s); RCCL %d reports ncclCommCount %d, rank %d, device %d"
               % (world, ci["rccl_version"], ci["rccl_nranks"], ci["rccl_rank"], ci["rccl_device"]),
               ci["rccl_nranks"] == world and ci["rccl_rank"] == rank)
    except Exception as e:
        report("ncclCommInitRank through cu2rec_comm_create", False, repr(e))
        sys.exit(1)
    tr, te = synth.make_ratings(4000 * world, 300, 60000 * world, min_degree=3, seed=5)
    f, hyper = 32, (0.01, 0.02, 0.02, 0.02, 0.02)
    u0, u1, my_tr, my_te = sharded.shard_of(tr, te, rank, world)
    P0 = cu.initialize_normal_array(tr.rows * f, f).reshape(tr.rows, f)
    ub0 = cu.initialize_normal_array(tr.rows, f)
    prev = cu.api.blocksolve_min_rate(8.0)  # (chains long enough for the block-solve kernels on this small set)
    try:
        model = cu.Model(u1 - u0, tr.cols, f, tr.global_bias, P=P0[u0:u1], user_bias=ub0[u0:u1])
        d_tr, d_te = cu.DeviceCSR(my_tr), cu.DeviceCSR(my_te)
        job = sharded.ShardJob(comm, model, d_tr, user_offset=u0, sync_every=10, merge="adaptive")
        job.run(hyper, 42, 0, 20, mode="blocksolve")
        loss = job.loss(d_te)
        info = job.info()
    finally:
        cu.api.blocksolve_min_rate(prev if prev > 0 else -1.0)
    report("two periods of sharded block-solve SGD, %d exchanges of %d bytes over RCCL" % (info["exchanges"], info["wire_bytes"]),
           info["exchanges"] == 2 and np.isfinite(loss["rmse"]))
    _, Q, _, ib = model.download()
    digest = hashlib.sha256(Q.tobytes() + ib.tobytes()).hexdigest() + " %.9g" % loss["rmse"]
    gathered = [None] * world
    dist.all_gather_object(gathered, digest)
    report("replicas of Q / item_bias bit-identical and the global loss equal on all ranks", len(set(gathered)) == 1, "" if len(set(gathered)) == 1 else str(gathered))
    job.close()
    comm.close()
    dist.barrier(device_ids=[local])
    dist.destroy_process_group()
    if rank == 0:
        with tempfile.TemporaryDirectory() as td:
            synth.write_csv(os.path.join(td, "train.csv"), tr)
            synth.write_csv(os.path.join(td, "test.csv"), te)
            with open(os.path.join(td, "c.cfg"), "w") as fh:
                fh.write("0 30 16 0.01 42 0.02 0.02 0.02 0.02\n")
            res = subprocess.run([os.path.join(ROOT, "bin", "mf"), "-c", os.path.join(td, "c.cfg"), "-g", str(world), os.path.join(td, "train.csv"),
                                  os.path.join(td, "test.csv")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
            files = all(os.path.exists(os.path.join(td, "train_f16_%s.csv" % c)) for c in ("p", "q", "user_bias", "item_bias", "global_bias"))
            report("bin/mf -g %d (forked ranks, id through pipes): exit code %d, five output files" % (world, res.returncode),
                   res.returncode == 0 and files and "TEST: Iteration 30" in res.stdout, "" if res.returncode == 0 else res.stdout[-800:])
        print("PREFLIGHT %s" % ("OK" if ok else "FAILED"), flush=True)
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
