#!/usr/bin/env python3
"""What one rank pays locally for one exchange of the sharded driver (cu2rec_shard_job_exchange): pack the item deltas
into the unpadded wire buffer, ncclAllReduce (a ONE-rank RCCL communicator here, i.e. the library call and its kernel
without any link traffic), apply.  The link time of a real N-rank ring comes on top.
  CU2REC_RCCL_WORLD1=1 python tools/exchange_probe.py [--workload ml-20m --factors 100 --merge weighted]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="ml-20m")
    ap.add_argument("--factors", type=int, default=100)
    ap.add_argument("--merge", default="weighted")
    ap.add_argument("--reps", type=int, default=50)
    args = ap.parse_args()
    os.environ.setdefault("CU2REC_RCCL_WORLD1", "1")
    import torch
    import bench
    import cu2rec_amd as cu
    from cu2rec_amd import sharded
    tr, _ = bench.load_dataset(args.workload, 20240917, 0, lambda: None)
    comm = sharded.Comm(0, 1)
    model = cu.Model(tr.rows, tr.cols, args.factors, tr.global_bias)
    d_tr = cu.DeviceCSR(tr)
    job = sharded.ShardJob(comm, model, d_tr, sync_every=1 << 30, merge=args.merge)
    job.run((0.01, 0.02, 0.02, 0.02, 0.02), 42, 0, 4, mode="hogwild")
    cu.lib().cu2rec_hogwild_resident(0)  # one-iteration calls: the streaming kernel either way
    hyper, it = (0.01, 0.02, 0.02, 0.02, 0.02), 4

    def loop(with_exchange):  # an exchange with nothing to send is skipped by the driver: one iteration in front of each
        nonlocal it
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.reps):
            job.run(hyper, 42, it, 1, mode="hogwild")
            it += 1
            if with_exchange:
                job.exchange()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / args.reps

    loop(True)
    dt = loop(True) - loop(False)
    info = job.info()
    print("%s f=%d merge=%s: wire buffer %d bytes; pack + one-rank ncclAllReduce + apply = %.1f us per exchange" %
          (args.workload, args.factors, args.merge, info["wire_bytes"], 1e6 * dt))


if __name__ == "__main__":
    main()
