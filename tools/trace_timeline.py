#!/usr/bin/env python3
"""Print the kernel timeline (start, end, duration in us; queue, stream, grid) of the last iterations in a
rocprofv3 --kernel-trace database (rocpd sqlite), aligned at the start of a bs_gram_kernel launch.

    python tools/trace_timeline.py <results.db> [n_iterations_from_end=3] [n_iterations_shown=2]
"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    shown = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    c = sqlite3.connect(db)
    rows = list(c.execute("select name,start,end,queue_id,stream_id,grid_x,workgroup_x from kernels order by start"))
    idx = [i for i, r in enumerate(rows) if "bs_gram" in r[0]]
    if len(idx) < back + 1:
        print("too few iterations in the trace")
        return
    i0 = idx[-back]
    i1 = idx[min(-back + shown, -1)]
    t0 = rows[i0][1]
    for r in rows[max(i0 - 3, 0):i1 + 1]:
        name = r[0].replace("cu2rec::(anonymous namespace)::", "").replace("void ", "")
        print(f"{(r[1]-t0)/1e3:9.1f} {(r[2]-t0)/1e3:9.1f} {(r[2]-r[1])/1e3:7.1f} q{r[3]} s{r[4]} wg{r[5]//max(r[6],1):5d} {name[:48]}")
    g = [rows[i][1] for i in idx]
    per = [(b - a) / 1e3 for a, b in zip(g, g[1:])]
    per.sort()
    print("gram-to-gram period us: median %.1f min %.1f (n=%d)" % (per[len(per)//2], per[0], len(per)))


if __name__ == "__main__":
    main()
