"""Per-kernel summary of a rocprofv3 rocpd database (rocprofv3 --kernel-trace --stats -d DIR -o NAME).
Usage: python scripts/rocpd_stats.py DB [out.csv]"""
import sqlite3
import sys


def stats(path):
    cur = sqlite3.connect(path).cursor()
    q = ("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels "
         "group by name order by sum(end-start) desc")
    rows = list(cur.execute(q))
    tot = sum(r[2] for r in rows) or 1
    return [(r[0], r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] / tot) for r in rows]


if __name__ == "__main__":
    rows = stats(sys.argv[1])
    lines = ['"Name","Calls","TotalDurationNs","AverageNs","MinNs","MaxNs","Percentage"']
    for r in rows:
        lines.append('"%s",%d,%d,%.1f,%d,%d,%.2f' % r)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write("\n".join(lines) + "\n")
    for r in rows[:15]:
        print("%-100s calls=%4d avg=%9.2f us  min=%9.2f  max=%9.2f  %5.1f%%" % (r[0][:100], r[1], r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, r[6]))
