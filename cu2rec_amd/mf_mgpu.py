"""bin/mf across the GPUs of one node: one process per GPU over RCCL.

  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
      -m cu2rec_amd.mf_mgpu -c ml.cfg [-m blocksolve|ordered|hogwild|serial] [--sync-every K]
      [--merge adaptive|mean|weighted|sum] train.csv test.csv   (defaults: blocksolve, adaptive -- as bin/mf)

Same stdout lines and the same five output CSVs as bin/mf (mf.cu:16-99); rank 0 prints and writes.
A thin launcher: ratings are sharded by user and trained by the C++ driver (cu2rec_amd/csrc/sharded.cpp,
cu2rec_train_sharded: ncclAllReduce of the item deltas each period); this script only starts the ranks' library calls,
hands rank 0's ncclUniqueId around and gathers the user side for the output files.  `bin/mf -g N` is the same thing
without Python.
"""
import argparse
import os
import sys

import numpy as np


def main(argv=None):
    ap = argparse.ArgumentParser(prog="cu2rec_amd.mf_mgpu")
    ap.add_argument("-c", dest="config", default=None)
    ap.add_argument("-m", dest="mode", default=None, choices=["hogwild", "ordered", "blocksolve", "serial"])
    ap.add_argument("--sync-every", type=int, default=0)
    ap.add_argument("--merge", default="adaptive", choices=["mean", "sum", "weighted", "adaptive"])
    ap.add_argument("train")
    ap.add_argument("test")
    args = ap.parse_args(argv)

    import torch
    import torch.distributed as dist

    import cu2rec_amd as cu
    from cu2rec_amd.sharded import Comm, share_through_torch, train_sharded

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        sys.exit("cu2rec_amd.mf_mgpu needs GPUs: there is no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    train, test = cu.createSparseMatrix(args.train), cu.createSparseMatrix(args.test)
    if test.rows > train.rows or test.cols > train.cols:
        sys.exit("the test file names users / items the training file does not have")
    cfg = cu.read_config(args.config) if args.config else cu.default_config()
    if rank == 0:
        cu.print_config(cfg)
    comm = Comm(rank, world, share=share_through_torch(device) if world > 1 else None)
    P, Q, losses, ub, ib, (u0, u1), _ = train_sharded(comm, train, test, cfg, mode=args.mode or cu.api.default_mode(cfg.n_factors), sync_every=args.sync_every,
                                                      merge=args.merge, verbose=True)
    # gather the user side on rank 0 (rows are contiguous per rank)
    if world > 1:
        parts_P, parts_ub = [None] * world, [None] * world
        dist.all_gather_object(parts_P, P)
        dist.all_gather_object(parts_ub, ub)
        P, ub = np.concatenate(parts_P, axis=0), np.concatenate(parts_ub)
    if rank == 0:
        parent = os.path.dirname(args.train) or "."
        base = os.path.splitext(os.path.basename(args.train))[0]
        f = cfg.n_factors
        cu.writeToFile(parent, base, "p", P, train.rows, f, f)  # mf.cu:83-87
        cu.writeToFile(parent, base, "q", Q, train.cols, f, f)
        cu.writeToFile(parent, base, "user_bias", ub, train.rows, 1, f)
        cu.writeToFile(parent, base, "item_bias", ib, train.cols, 1, f)
        cu.writeToFile(parent, base, "global_bias", np.array([train.global_bias], np.float32), 1, 1, f)
    comm.close()
    if world > 1:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
