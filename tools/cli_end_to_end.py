#!/usr/bin/env python3
"""End-to-end wall time of bin/mf on an ML-20M-shape CSV pair (parse, upload, train, download, write the five CSVs)."""
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def fast_write(path, csr):
    user = np.repeat(np.arange(csr.rows), np.diff(csr.indptr)) + 1
    with open(path, "w") as fh:
        fh.write("userId,itemId,rating\n")
        step = 2_000_000
        for s in range(0, csr.nnz, step):
            e = min(csr.nnz, s + step)
            u, i, r = user[s:e].astype(str), (csr.indices[s:e] + 1).astype(str), csr.data[s:e].astype(str)
            fh.write("\n".join(np.char.add(np.char.add(np.char.add(u, ","), np.char.add(i, ",")), r).tolist()) + "\n")


def main():
    import bench
    mode = sys.argv[1] if len(sys.argv) > 1 else "hogwild"
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    tr, te = bench.load_dataset("ml-20m", 20240917, 0, lambda: None)
    td = tempfile.mkdtemp(prefix="cu2rec_cli_")
    ptr, pte, pcfg = os.path.join(td, "train.csv"), os.path.join(td, "test.csv"), os.path.join(td, "c.cfg")
    t0 = time.perf_counter()
    fast_write(ptr, tr)
    fast_write(pte, te)
    print("wrote CSVs (%.0f MB + %.0f MB) in %.1f s" % (os.path.getsize(ptr) / 1e6, os.path.getsize(pte) / 1e6, time.perf_counter() - t0))
    with open(pcfg, "w") as fh:
        fh.write("0 %d 100 0.01 42 0.02 0.02 0.02 0.02\n" % iters)
    t0 = time.perf_counter()
    out = subprocess.run([os.path.join(ROOT, "bin", "mf"), "-c", pcfg, "-m", mode, ptr, pte], stdout=subprocess.PIPE, text=True, check=True).stdout
    wall = time.perf_counter() - t0
    lines = [l for l in out.split("\n") if l.startswith(("TRAIN", "TEST", "Time taken"))]
    print("\n".join(lines))
    print("bin/mf -m %s, %d iterations, f=100: wall %.2f s end to end; outputs: %s" % (
        mode, iters, wall, ", ".join("%s %.0f MB" % (n, os.path.getsize(os.path.join(td, n)) / 1e6) for n in sorted(os.listdir(td)) if "_f100_" in n)))


if __name__ == "__main__":
    main()
