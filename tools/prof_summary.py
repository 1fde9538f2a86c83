#!/usr/bin/env python3
"""Per-kernel summary of a `rocprofv3 --kernel-trace` run from its rocpd database (rocprofv3 of ROCm 7.2 writes
<name>_results.db): calls, total / average / min / max duration, share of the total -- the table `--stats` prints, as CSV.

    python tools/prof_summary.py gpurun_out/<dir>/<name>_results.db > profiles/<name>_kernel_stats.csv
"""
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    rows = c.execute("select name, count(*), sum(end - start), avg(end - start), min(end - start), max(end - start) "
                     "from kernels group by name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"')
    for name, n, tot, avg, mn, mx in rows:
        print('"%s",%d,%d,%.3f,%.4f,%d,%d' % (name, n, tot, avg, 100.0 * tot / total, mn, mx))


if __name__ == "__main__":
    main(sys.argv[1])
