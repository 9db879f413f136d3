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
Prints one PASS / FAIL line per step; exit code 0 only if all passed.  No performance number comes out of this tool."""
import argparse
import hashlib
import os
import socket
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    args = ap.parse_args()
    if "WORLD_SIZE" not in os.environ:
        if args.gpus == 1:
            os.environ["CU2REC_RCCL_WORLD1"] = "1"
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__), "--gpus", str(args.gpus)]
        sys.exit(subprocess.run(cmd).returncode)  # fresh child processes: nothing that touched a GPU is re-executed

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
        report("ncclCommInitRank through cu2rec_comm_create, %d rank(s)" % world, True)
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
