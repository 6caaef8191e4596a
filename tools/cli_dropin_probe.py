#!/usr/bin/env python3
"""Where the time of the drop-in command line goes on bench.py's CPU-baseline sample (120 k read pairs): every plass-hip invocation with
PLASSHIP_POOL_STATS=1, its own "Time for processing", the wall clock around the process — once while this process still holds its
context (as bench.py's cpu_baseline leg runs them) and once after closing it."""
import os, subprocess, sys, tempfile, time, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, plass_amd, __graft_entry__ as g

def chain(td, tag):
    hip = os.path.join(ROOT, "plass_amd", "plass-hip")
    env = dict(g.child_env(), PLASSHIP_POOL_STATS="1")
    tot = 0.0
    for it in range(3):
        s, p, a, o = (os.path.join(td, x) for x in ("seq_%d" % it, "pref", "aln", "seq_%d" % (it + 1)))
        km = ["--alph-size", "13", "--kmer-per-seq", "60", "--kmer-per-seq-scale", "0", "-k", "14", "-c", "0", "--hash-shift", str(bench.hash_shift(it)),
              "--include-only-extendable", "1" if it else "0", "--ignore-multi-kmer", "1"]
        rs = ["--rescore-mode", "3", "--min-seq-id", "0.9", "-e", "1e-5", "-c", "0"]
        asm = ["--min-seq-id", "0.9", "--max-seq-len", "65535", "--keep-target", "1", "--rescore-mode", "3"]
        for args in (["kmermatcher", s, p] + km, ["rescorediagonal", s, s, p, a] + rs, ["assembleresults", s, a, o] + asm):
            t0 = time.time()
            out = subprocess.run([hip] + args, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env)
            wall = time.time() - t0
            m = re.search(r"Time for processing: ([0-9.]+)s", out.stdout)
            pool = [l for l in out.stdout.splitlines() if "pool" in l]
            tot += float(m.group(1)) if m else 0.0
            print("%s it%d %-16s processing %s s, wall %.3f s | %s" % (tag, it, args[0], m.group(1) if m else "?", wall, " ".join(pool)[:200]), flush=True)
    print("%s: sum of processing %.3f s" % (tag, tot), flush=True)

ctx = plass_amd.Context()
db, desc = bench.build_workload(ctx, "c3", 120000, min_genomes=5)
with tempfile.TemporaryDirectory() as td:
    db.write(os.path.join(td, "seq_0")); db.free()
    chain(td, "parent context open")
    # what bench.py holds at that point: a large arena
    big = ctx.read_seqdb(os.path.join(td, "seq_0"))
    chain(td, "parent context open, again")
    big.free()
    ctx.close()
    chain(td, "parent context closed")
    for k in ("PLASSHIP_HOST_THREADS",):
        os.environ[k] = "16"
    chain(td, "16 host threads")
