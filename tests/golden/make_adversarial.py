#!/usr/bin/env python3
"""GENERATION-TIME ONLY (build container): hostile protein and nucleotide inputs for tests/golden/adversarial.tar.gz.
Writes aa_seq / nucl_seq (index order != key order, sparse keys) into the directory given as argv[1]; the reference
binaries are then run on them exactly like in make_golden.sh (commands in the fixture's MANIFEST)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from plass_amd import synth


def main(out):
    rng = np.random.default_rng(5)
    aa = "ACDEFGHIKLMNPQRSTVWY"
    base = "".join(rng.choice(list(aa), size=4000))
    seqs = []
    for i in range(400):
        p = int(rng.integers(0, 3900)); l = int(rng.integers(40, 100))
        seqs.append(base[p:p + l])
    seqs += ["MKV", "A" * 13, "A" * 14, "A" * 200, "AS" * 120, "MKVLAAGX" * 10, "XXXXXXXXXXXXXXXXXXXXXXXX", base[100:180] + "*", base[100:180] + "*",
             base[1000:1100].lower(), base[1000:1100], "*" + base[2000:2060], base[500:2500]]
    long_contig = "".join(rng.choice(list(aa), size=33000))
    seqs += [long_contig, long_contig[32000:] + base[:60], base[3000:3050] + long_contig[:70]]
    keys = np.cumsum(rng.integers(1, 4, size=len(seqs))).astype(np.uint32)
    perm = rng.permutation(len(seqs))
    data, off, elen, key = synth.pack_db([np.frombuffer(seqs[i].encode(), dtype=np.uint8) for i in perm])
    synth.write_db(os.path.join(out, "aa_seq"), data, off, elen, keys[perm], 0)

    rng = np.random.default_rng(9)
    g = "".join(rng.choice(list("ACGT"), size=6000))
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    rc = lambda s: "".join(comp.get(c, "N") for c in reversed(s))
    ns = []
    for i in range(300):
        p = int(rng.integers(0, 5800)); l = int(rng.integers(60, 200)); s = g[p:p + l]
        if rng.random() < 0.5:
            s = rc(s)
        ns.append(s)
    ns += ["ACGT", "A" * 21, "A" * 22, "A" * 300, "AT" * 100, "ACGTACGTAC" * 20, "N" * 50, g[100:250].lower(), g[100:250], g[100:160] + "N" + g[161:250],
           g[300:400] + "RYKM" + g[404:500], g[1000:1150], rc(g[1000:1150]), g[1000:1150], "ACGTNNNNACGT" * 10, g[2000:2200].replace("A", "a"),
           g[3000:3100] + "U" + g[3101:3200], g[500:3500]]
    nkeys = np.cumsum(rng.integers(1, 4, size=len(ns))).astype(np.uint32)
    nperm = rng.permutation(len(ns))
    data, off, elen, key = synth.pack_db([np.frombuffer(ns[i].encode(), dtype=np.uint8) for i in nperm])
    synth.write_db(os.path.join(out, "nucl_seq"), data, off, elen, nkeys[nperm], 1)


if __name__ == "__main__":
    main(sys.argv[1])
