#!/usr/bin/env python3
"""Host boundary throughput on the GPU box: DB files <-> HBM through the pinned staging buffers and the threaded text formatters /
parsers (plasship_seqdb_read/write, plasship_cands_read/write, plasship_alns_read/write), on the fragment DB of a C3-model read set.
    python tools/io_probe.py [pairs]     (PLASSHIP_HOST_THREADS=n to vary the host threads)"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, plass_amd

pairs = int(float(sys.argv[1])) if len(sys.argv) > 1 else 2500000
ctx = plass_amd.Context(0)
db, desc = bench.build_workload(ctx, "c3", pairs)
print("fragments:", desc["protein_fragments"], "residues:", desc["fragment_residues"], "host threads:", os.environ.get("PLASSHIP_HOST_THREADS", "default"))


def timed(what, f, nbytes=None):
    t0 = time.perf_counter(); r = f(); ctx.sync(); dt = time.perf_counter() - t0
    print("%-28s %7.3f s%s" % (what, dt, "  %6.2f GB/s" % (nbytes / dt / 1e9) if nbytes else ""), flush=True)
    return r


with tempfile.TemporaryDirectory() as td:
    P = lambda n: os.path.join(td, n)
    size = lambda n: os.path.getsize(P(n)) + os.path.getsize(P(n) + ".index")
    timed("seqdb_write", lambda: db.write(P("seq")))
    print("   files: %.1f MB" % (size("seq") / 1e6))
    db2 = timed("seqdb_read", lambda: ctx.read_seqdb(P("seq")), size("seq"))
    db2.free()
    out, kst, rst, ast, _ = None, None, None, None, None
    par = plass_amd.KmermatchParams(k=14, alph_size=13, kmer_per_seq=60, kmer_per_seq_scale=0.0, hash_shift=67, include_only_extendable=False, ignore_multi_kmer=True, cov_mode=0, c=0.0)
    cands, kst = timed("kmermatcher (GPU)", lambda: ctx.kmermatcher(db, par))
    timed("cands_write", lambda: cands.write(P("pref")))
    print("   files: %.1f MB, %d candidate lines" % (size("pref") / 1e6, cands.count()))
    c2 = timed("cands_read", lambda: ctx.read_prefdb(db, db, P("pref")), size("pref"))
    c2.free()
    alns, rst = timed("rescorediagonal (GPU)", lambda: ctx.rescorediagonal(db, db, cands, plass_amd.RescoreParams(min_seq_id=0.9, e=1e-5)))
    timed("alns_write", lambda: alns.write(P("aln")))
    print("   files: %.1f MB, %d alignment lines" % (size("aln") / 1e6, alns.count()))
    a2 = timed("alns_read", lambda: ctx.read_alndb(db, P("aln")), size("aln"))
    a2.free()
    o, ast = timed("assembleresults (GPU)", lambda: ctx.assembleresults(db, alns, plass_amd.AssembleParams(min_seq_id=0.9, max_seq_len=65535, keep_target=True)))
ctx.close()
