"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace) into a --stats style table."""
import sqlite3
import sys


def main(path, out=None):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    namec = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = list(cur.execute(f"select {namec}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by {namec} order by 3 desc"))
    tot = sum(r[2] for r in rows) or 1
    lines = ["%-70s %8s %14s %12s %12s %12s %7s" % ("KERNEL", "CALLS", "TOTAL_ns", "AVG_ns", "MIN_ns", "MAX_ns", "PCT")]
    for n, c, s, a, mn, mx in rows:
        lines.append("%-70s %8d %14d %12.0f %12d %12d %6.2f%%" % (n[:70], c, s, a, mn, mx, 100.0 * s / tot))
    txt = "\n".join(lines)
    print(txt)
    if out:
        open(out, "w").write(txt + "\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
