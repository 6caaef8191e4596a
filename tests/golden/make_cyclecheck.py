#!/usr/bin/env python3
"""GENERATION-TIME ONLY (build container): input DB of tests/golden/cyclecheck.tar.gz — circular genomes assembled past their
end (exact and with errors), linear controls, repeats, hostile and boundary cases, reads.  The expected outputs in the tarball
were written by the unmodified reference (`penguin cyclecheck`, see make_golden.sh)."""
import sys

import numpy as np


def sequences():
    rng = np.random.default_rng(5)
    B = "ACGT"

    def rnd(n):
        return "".join(B[i] for i in rng.integers(0, 4, n))

    def mutate(s, rate):
        a = list(s)
        for i in np.nonzero(rng.random(len(a)) < rate)[0]:
            a[i] = B[(B.index(a[i]) + 1 + int(rng.integers(0, 3))) % 4]
        return "".join(a)

    seqs = []
    for n in (300, 700, 1500, 2500, 5000, 9000, 20000):          # G + G[:x]
        for frac in (0.05, 0.2, 0.45, 0.9):
            g = rnd(n); x = max(30, int(n * frac))
            seqs.append(g + g[:x])
            seqs.append(g + mutate(g[:x], 0.02))
    for n in (200, 1000, 8000, 30000):                           # linear controls
        seqs.append(rnd(n))
    g = rnd(3000); seqs.append(g[:1000] + rnd(500) + g[:1000] + rnd(700))      # interspersed repeat
    u = rnd(37); seqs.append(u * 60)                                          # tandem repeat
    seqs.append("A" * 900); seqs.append("ACGT" * 300); seqs.append(rnd(400) + "N" * 300 + rnd(400))
    g = rnd(4000); seqs.append(g + "NNNNNNNNNN" + g[:1500])
    seqs += ["ACGTACGTAC", "A" * 22, rnd(23), rnd(65), rnd(66), rnd(67), ""]
    g = rnd(2000); seqs.append((g + g[:600]).lower())
    g = rnd(6000); seqs.append(g * 3)                                         # three full copies
    g = rnd(38000); seqs.append(g + g[:9000])                                 # long contig
    seqs.append(rnd(49999)); seqs.append(rnd(50000)); seqs.append(rnd(50010))  # below / at / above --max-seq-len 50000
    G = rnd(20000)
    for _ in range(400):                                                      # reads (most entries of a real DB)
        p = int(rng.integers(0, 20000 - 150)); seqs.append(G[p:p + 150])
    return seqs


if __name__ == "__main__":
    out = sys.argv[1]
    seqs = sequences()
    data = b""; idx = []
    for i, s in enumerate(seqs):
        e = s.encode() + b"\n\0"
        idx.append("%d\t%d\t%d\n" % (3 * i, len(data), len(e)))
        data += e
    open(out, "wb").write(data)
    open(out + ".index", "w").write("".join(idx))
    open(out + ".dbtype", "wb").write((1).to_bytes(4, "little"))
    print(len(seqs), "sequences,", sum(len(s) for s in seqs), "nt")
