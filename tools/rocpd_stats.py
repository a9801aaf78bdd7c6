"""Summarise a rocprofv3 rocpd SQLite database (kernel-trace) into a per-kernel table (calls, total, avg, %).

    python tools/rocpd_stats.py gpurun_out/prof/x_results.db [--skip-first N] > profiles/rNN_name_kernel_stats.txt
"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in c.execute("pragma table_info(%s)" % disp)]
    scol = [r[1] for r in c.execute("pragma table_info(%s)" % sym)]
    name_col = "kernel_name" if "kernel_name" in scol else "display_name"
    q = ("select s.%s, count(*), sum(d.end - d.start), min(d.end - d.start), max(d.end - d.start) "
         "from %s d join %s s on d.kernel_id = s.id group by s.%s order by 3 desc" % (name_col, disp, sym, name_col))
    rows = list(c.execute(q))
    tot = sum(r[2] for r in rows) or 1
    print("# %s" % db)
    print("# columns: %s" % ",".join(cols))
    print("%-90s %8s %12s %10s %10s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct"))
    for n, k, t, mn, mx in rows:
        short = n if len(n) <= 90 else n[:87] + "..."
        print("%-90s %8d %12.1f %10.2f %10.2f %10.2f %6.2f" % (short, k, t / 1e3, t / k / 1e3, mn / 1e3, mx / 1e3, 100.0 * t / tot))
    print("# total kernel time: %.3f ms" % (tot / 1e6))


if __name__ == "__main__":
    main()
