#!/bin/bash
# round 4, GPU call 14: kernel trace of configs[4] after the extension / cyclecheck rewrites
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_call14; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $R/$O/prof -o c5 -- python $R/bench.py --config c5 --steps 10 --warmup 0 --no-cpu-baseline --no-verify > $R/$O/bench_c5_rocprof.log 2> $R/$O/bench_c5_rocprof.err
cd $R
python tools/rocpd_summary.py $O/prof/c5_results.db > $O/kernel_stats_c5.txt 2>&1
python - $O/prof/c5_results.db > $O/kernel_last_iteration_c5.txt <<'PY'
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
rows = con.execute("select name, start, end from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if "boundsKernel" in r[0]]
s = idx[-1]
agg = {}
for name, st, en in rows[s:]:
    a = agg.setdefault(name, [0, 0]); a[0] += 1; a[1] += en - st
print("last step (iteration 9): %d launches, %.1f ms in kernels, %.1f ms from first start to last end" % (len(rows) - s, sum(v[1] for v in agg.values()) / 1e6, (rows[-1][2] - rows[s][1]) / 1e6))
for name, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print("%-110s %5d %10.3f ms" % (name[:110], c, t / 1e6))
PY
head -30 $O/kernel_last_iteration_c5.txt | cut -c1-140
find $O -name "*.db" -size +30M -delete
