#!/usr/bin/env python3
"""GPU box probe (round 6): the distribution of window counts (L - k + 1, k = 14) over the sequences of every iteration of the 50 M-read chain, with the
extraction tier each class goes to — to cost a thread-per-sequence tier for sequences a little longer than reads, or sub-wavefront groups in the 4-scores tier
(VERDICT r5 item 1), before building either."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, plass_amd
from plass_amd import _lib
pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 25000000
ctx = plass_amd.Context(0)
db, wl = bench.build_workload(ctx, "c3", pairs)
edges = [0, 59, 64, 75, 96, 128, 192, 256, 512, 1024, 3072, 1 << 30]
print("windows per sequence (k = 14): " + " | ".join("<=%d" % e for e in edges[1:-1]) + " | more")
for it in range(12):
    n = db.info()["n"]
    elen = np.zeros(n, dtype=np.uint32)
    _lib._check(ctx.lib.plasship_seqdb_download(ctx.h, db.h, None, None, elen.ctypes.data, None), "plasship_seqdb_download")
    nwin = np.maximum(elen.astype(np.int64) - 2 - 13, 0)
    cnt, _ = np.histogram(nwin, bins=[e + 0.5 for e in [-1] + edges[1:]])
    res, _ = np.histogram(nwin, bins=[e + 0.5 for e in [-1] + edges[1:]], weights=nwin)
    print("iteration %2d: sequences (M) %s" % (it, " ".join("%7.2f" % (c / 1e6) for c in cnt)))
    print("              windows  (G) %s" % " ".join("%7.3f" % (r / 1e9) for r in res), flush=True)
    out, kst, rst, ast, wall = bench.one_iteration(ctx, db, it)
    print("              extraction %.1f ms (thread-per-sequence %.1f, wave tiers + cached %.1f), cached sequences %.2f M" % (kst.ms_extract, kst.ms_extract_short_kernel, kst.ms_extract_wave_kernel, kst.n_cached_sequences / 1e6), flush=True)
    db.free(); db = out
