#!/usr/bin/env python3
"""Per-kernel sums of the counters of one rocprofv3 --pmc pass (rocpd database, `counters_collection` view).
Usage: tools/rocpd_pmc.py gpurun_out/pmc/x_results.db [N_KERNELS]"""
import collections
import sqlite3
import sys


def main():
    con = sqlite3.connect(sys.argv[1])
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set); dur = collections.defaultdict(float)
    for k, c, v, d, du in con.execute("select kernel_name, counter_name, value, dispatch_id, duration from counters_collection"):
        agg[k][c] += v
        if d not in disp[k]:
            disp[k].add(d); dur[k] += du
    rows = sorted(agg.items(), key=lambda kv: -dur[kv[0]])
    print("# per dispatch averages; SQ_WAVE_CYCLES / SQ_WAIT_* count quad-cycles (MI355X_MICROARCH.md)")
    for k, c in rows[:top]:
        n = len(disp[k])
        print(k[:110])
        print("    dispatches=%d avg_us=%.1f " % (n, dur[k] / n / 1e3) + " ".join("%s=%.4g" % (x, c[x] / n) for x in sorted(c)))


if __name__ == "__main__":
    main()
