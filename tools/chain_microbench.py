#!/usr/bin/env python3
"""What a two-wave chain of the ordered kernel costs per link, and how many of them a chip runs at once: N items x L users each, every
user rates exactly ONE item (so every iteration every item collects exactly L updates: N chains of L links, each chain's users at random
rows), ordered mode (no block solves).  Per (N, L, f): microseconds per iteration, ns per link of one
chain (us / L), links per microsecond of the chip.
usage: tools/chain_microbench.py [--factors 100] [--links 256] [--chains 1,64,256,512,1024,2048,4096] [--mode ordered|blocksolve]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cu2rec_amd as cu
from cu2rec_amd.api import HostCSR

ap = argparse.ArgumentParser()
ap.add_argument("--factors", type=int, default=100)
ap.add_argument("--links", type=int, default=256)
ap.add_argument("--chains", default="1,64,256,512,1024,2048,4096")
ap.add_argument("--mode", default="ordered")
ap.add_argument("--iters", type=int, default=200)
ap.add_argument("--layout", default="random", choices=["random", "contiguous", "strided"])
args = ap.parse_args()
cu.api.blocksolve_min_rate(1e9)  # (no chain is solved block-wise: the two-wave form for all of them)
hyper = (0.01, 0.02, 0.02, 0.02, 0.02)
for n in (int(v) for v in args.chains.split(",")):
    L, f = args.links, args.factors
    users = n * L
    if args.layout == "random":      # L users per item, at random rows
        items = np.random.RandomState(n).permutation(np.arange(users, dtype=np.int32) % n).astype(np.int32)
    elif args.layout == "contiguous":  # chain y = users [y L, (y + 1) L)
        items = (np.arange(users, dtype=np.int32) // L).astype(np.int32)
    else:                              # chain y = users y, y + n, ...
        items = (np.arange(users, dtype=np.int32) % n).astype(np.int32)
    tr = HostCSR(np.arange(users + 1, dtype=np.int32), items, np.full(users, 3.0, np.float32), users, n + 2, 3.0)  # (two items nobody rates: the key's rank field must hold n + 1, or the last chain is left to the one-group walk)
    d = cu.DeviceCSR(tr)
    m = cu.Model(users, n + 2, f, 3.0)
    m.sgd(d, hyper, 42, 0, 70, mode=args.mode)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        m.sgd(d, hyper, 42, 70 + rep * args.iters, args.iters, mode=args.mode)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / args.iters)
    print("%s f=%d chains %5d x %4d links: %8.2f us per iteration  = %6.1f ns per link of a chain, %7.1f links/us chip-wide"
          % (args.layout, f, n, L, 1e6 * best, 1e9 * best / L, users / (1e6 * best)), flush=True)
