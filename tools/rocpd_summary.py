#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (*_results.db): per-kernel count / total / average from the `kernels` view, and
(with --timeline N) the launch sequence of the N-th-from-last assembly iteration with the idle gaps between kernels.
Usage: tools/rocpd_summary.py gpurun_out/prof/x_results.db [--timeline 1]"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    con = sqlite3.connect(db)
    rows = con.execute("select name, start, end from kernels order by start").fetchall()
    agg = {}
    for name, st, en in rows:
        a = agg.setdefault(name, [0, 0])
        a[0] += 1; a[1] += en - st
    total = sum(v[1] for v in agg.values())
    print("%-100s %8s %12s %12s %6s" % ("kernel", "calls", "total_ms", "avg_us", "%"))
    for name, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("%-100s %8d %12.3f %12.2f %6.2f" % (name[:100], c, t / 1e6, t / c / 1e3, 100.0 * t / total))
    if "--timeline" in sys.argv:
        back = int(sys.argv[sys.argv.index("--timeline") + 1])
        idx = [i for i, r in enumerate(rows) if "boundsKernel" in r[0]]
        s, e = idx[-1 - back], (idx[-back] if back else len(rows))
        prev = None; gaps = 0.0; kern = 0.0
        print("\niteration timeline (gap before launch | duration | kernel)")
        for name, st, en in rows[s:e]:
            gap = (st - prev) / 1e3 if prev else 0.0
            gaps += max(gap, 0.0); kern += (en - st) / 1e3
            print("%9.1f us | %9.1f us | %s" % (gap, (en - st) / 1e3, name[:90]))
            prev = en
        print("idle between kernels %.1f us, kernels %.1f us" % (gaps, kern))


if __name__ == "__main__":
    main()
