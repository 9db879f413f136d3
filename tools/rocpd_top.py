#!/usr/bin/env python3
"""Per-kernel summary (calls, total / average / min / max us) of a rocprofv3 rocpd database (ROCm 7.2 writes
<name>_results.db by default).  usage: rocpd_top.py results.db [--csv out.csv]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    rows = cur.execute("select name, count(*), sum(end - start), avg(end - start), min(end - start), max(end - start) "
                       "from kernels group by name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    lines = ["name,calls,total_us,avg_us,min_us,max_us,percent"]
    for name, calls, tot, avg, mn, mx in rows:
        lines.append('"%s",%d,%.3f,%.3f,%.3f,%.3f,%.2f' % (name[:110], calls, tot / 1e3, avg / 1e3, mn / 1e3, mx / 1e3, 100.0 * tot / total))
    out = "\n".join(lines) + "\n"
    if "--csv" in sys.argv:
        open(sys.argv[sys.argv.index("--csv") + 1], "w").write(out)
    sys.stdout.write(out)


if __name__ == "__main__":
    main()
