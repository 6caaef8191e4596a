#!/usr/bin/env python3
"""Timing probe for the nucleotide path (PenguiN's nuclassemble stage: kmermatcher k=22 -> rescorediagonal ->
nuclassembleresults -> cyclecheck) on synthetic 2x150 nt reads, chained on the device.  Not the graded benchmark (bench.py is);
prints per-iteration stage times.  Usage: tools/bench_nucl.py [read_pairs] [iterations]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import plass_amd
from plass_amd import synth


def main():
    pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    data, off, elen, key = synth.nucleotide_read_db(pairs, seed=7)
    ctx = plass_amd.Context(0)
    db0 = ctx.upload_seqdb(data, off, elen, key, 1)
    kp = plass_amd.KmermatchParams(k=22, alph_size=5, kmer_per_seq=60, kmer_per_seq_scale=0.1, hash_shift=67, include_only_extendable=True,
                                   ignore_multi_kmer=True, cov_mode=0, c=0.0)
    for rep in range(2):                                   # first chain warms the allocator
        db = db0
        for it in range(iters):
            ctx.sync(); t0 = time.perf_counter()
            cands, kst = ctx.kmermatcher(db, kp)
            ctx.sync(); t1 = time.perf_counter()
            alns, rst = ctx.rescorediagonal(db, db, cands, plass_amd.RescoreParams(min_seq_id=0.99))
            ctx.sync(); t2 = time.perf_counter()
            out, ast = ctx.assembleresults(db, alns, plass_amd.AssembleParams(min_seq_id=0.99, max_seq_len=200000))
            ctx.sync(); t3 = time.perf_counter()
            cyc, rest, cst = ctx.cyclecheck(out, max_seq_len=200000, chop_cycle=True, with_rest=True)      # data/nuclassemble.sh:132
            ctx.sync(); t4 = time.perf_counter()
            if rep:
                print("it%d: kmermatcher %.2f ms (extract %.2f partition %.2f group %.2f sort2 %.2f reduce %.2f; Nk=%d Nm=%d Nc=%d) | rescore %.2f ms (kernel %.2f) | "
                      "nuclassemble %.2f ms (kernel %.2f, extended %d, rescored %d) | cyclecheck %.2f ms (circular %d; tiers %d/%d/%d)" % (
                          it, (t1 - t0) * 1e3, kst.ms_extract, kst.ms_sort1, kst.ms_group, kst.ms_sort2, kst.ms_reduce, kst.n_kmer_records, kst.n_grouped,
                          kst.n_candidates, (t2 - t1) * 1e3, rst.ms_kernel, (t3 - t2) * 1e3, ast.ms_tier_kernel[0], ast.n_extended, ast.n_rescored,
                          (t4 - t3) * 1e3, cst.n_cyclic, cst.n_wave_small, cst.n_wave_large, cst.n_block))
            alns.free(); cands.free(); cyc.free(); out.free(); out = rest
            if db is not db0:
                db.free()
            db = out


if __name__ == "__main__":
    main()
