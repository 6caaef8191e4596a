#!/usr/bin/env python3
"""HBM traffic per kernel from two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; rocpd databases).
Usage: tools/rocpd_traffic.py FETCH_results.db WRITE_results.db [N_KERNELS [OUT.json [STEPS [COMMAND]]]]
OUT.json (profiles/r03_pmc_traffic.json) is what bench.py reads `roofline.traffic` from: per kernel, HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE.
STEPS = assembly iterations the profiled command ran (warm-up, timed and verification iterations together); the file records the hash
of the product sources (bench.source_sha) and bench.py refuses it for any other code."""
import collections
import json
import sqlite3
import sys


def load(db, name):
    con = sqlite3.connect(db); agg = collections.defaultdict(float); disp = collections.defaultdict(set)
    for k, c, v, d in con.execute("select kernel_name, counter_name, value, dispatch_id from counters_collection"):
        if c == name:
            agg[k] += v; disp[k].add(d)
    return {k: (agg[k] / len(disp[k]), len(disp[k])) for k in agg}


def main():
    f = load(sys.argv[1], "FETCH_SIZE"); w = load(sys.argv[2], "WRITE_SIZE")
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 24
    rows = sorted(f, key=lambda k: -(f[k][0] + w.get(k, (0, 0))[0]))
    print("# counter values are KB per dispatch (averaged over the dispatches of each kernel).  gfx950 note (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports")
    print("# 1/2 of the bytes of wide coalesced streaming reads (16 B / lane) -> the corrected column doubles it; WRITE_SIZE is uncalibrated.")
    print("%-92s %8s %14s %14s %14s" % ("kernel", "launches", "FETCH MB", "FETCHx2 MB", "WRITE MB"))
    for k in rows[:top]:
        print("%-92s %8d %14.1f %14.1f %14.1f" % (k[:92], f[k][1], f[k][0] / 1024, 2 * f[k][0] / 1024, w.get(k, (0, 0))[0] / 1024))


    if len(sys.argv) > 4:
        import os
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        cmd = sys.argv[6] if len(sys.argv) > 6 else "python bench.py --gpus 1 --steps 20 --warmup 5"
        out = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (two passes) of `%s`" % cmd,
               "source_sha": bench.source_sha(),
               "steps": int(sys.argv[5]) if len(sys.argv) > 5 else 12,
               "correction": "FETCH_SIZE doubled (gfx950: 128-B requests tallied at 64 B, MI355X_MICROARCH.md); WRITE_SIZE as reported; counter unit KB",
               "kernels": {k: {"launches": f[k][1], "fetch_bytes_per_launch": 2 * f[k][0] * 1024, "write_bytes_per_launch": w.get(k, (0, 0))[0] * 1024,
                               "hbm_bytes_per_launch": 2 * f[k][0] * 1024 + w.get(k, (0, 0))[0] * 1024} for k in rows}}
        json.dump(out, open(sys.argv[4], "w"), indent=1)


if __name__ == "__main__":
    main()
