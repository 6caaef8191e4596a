#!/usr/bin/env python3
"""Digests of the first iterations of a protein chain on a synthetic read set, one implementation path per process — test infrastructure.

    chain_digests.py <config> <read pairs> <iterations> <mode>       mode: lines | legacy | sharded1

`lines` is the single-GPU path (line-store partition), `legacy` the dense histogram + scatter partition and the three-phase group
kernel (PLASSHIP_LEGACY_PARTITION=1, read once per process: hence one process per mode), `sharded1` the sharded orchestration
(owner partition, exchanges, halo, merge of extended sequences) in a 1-rank group.  Prints one JSON line: per iteration the counts
and the order-independent digest (include/plasship.h: plasship_seqdb_digest) of seq_{i+1}.  tests/test_gpu_large.py compares the
three: independent implementations agreeing where no CPU oracle can follow (5.3 G record slots at 50 M reads)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    cfg, pairs, iters, mode = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    if mode == "legacy":
        os.environ["PLASSHIP_LEGACY_PARTITION"] = "1"
    if mode == "sharded1":
        # ONE rank holds the send and receive buffers eight ranks would share (three record arrays of 60 GB at 50 M reads): it gets a
        # larger share of the HBM than the 88 % default (nothing else runs in this process)
        os.environ.setdefault("PLASSHIP_POOL_FRACTION", "0.96")
    import bench
    import plass_amd
    ctx = plass_amd.Context(0)
    db, wl = bench.build_workload(ctx, cfg, pairs)
    res = {"mode": mode, "fragments": wl["protein_fragments"], "fragments_digest": db.digest()[0], "iterations": []}

    def chain(rank, ctx):
        d = db
        for it in range(iters):
            out, kst, rst, ast, _ = bench.one_iteration(ctx, d, it)
            res["iterations"].append({"N_k": kst.n_kmer_records, "N_m": kst.n_grouped, "N_c": kst.n_candidates, "verified": rst.n_accepted,
                                      "extended": ast.n_extended, "digest": out.digest()[0], "residues": out.info()["residues"]})
            if d is not db:
                d.free()
            d = out
        if d is not db:
            d.free()

    if mode == "sharded1":
        from plass_amd.shard import LocalGroup
        LocalGroup(1).run(chain, [ctx])
    else:
        chain(0, ctx)
    db.free()
    ctx.close()
    print("CHAIN_DIGESTS " + json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
