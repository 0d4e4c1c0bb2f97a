"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / avg / min / max (us).
Usage: python tools/rocpd_stats.py <results.db> [out.csv]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = list(cur.execute(
        "select %s, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels "
        "group by %s order by sum(end-start) desc" % (name_col, name_col)))
    total = sum(r[2] for r in rows) or 1
    lines = ["Name,Calls,TotalDurationUs,AverageUs,MinUs,MaxUs,Percentage"]
    for n, c, s, a, mn, mx in rows:
        lines.append('"%s",%d,%.1f,%.2f,%.2f,%.2f,%.2f' % (n, c, s / 1e3, a / 1e3, mn / 1e3, mx / 1e3, 100.0 * s / total))
    out = "\n".join(lines)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out + "\n")
    print(out)


if __name__ == "__main__":
    main()
