#!/usr/bin/env python3
"""GENERATION-TIME ONLY (build container): input DB of tests/golden/orfs.tar.gz (row N2: extractorfs / translatenucs) — random
reads with N runs, IUPAC ambiguity codes, lower case, U/u, illegal characters, sequences shorter than a codon, hand-made ORF
layouts, plus plain 150-nt reads.  The expected outputs in the tarball were written by the unmodified reference
(`plass extractorfs` with the flag lines of orf/FLAGS, `plass translatenucs --add-orf-stop 1 / 0`)."""
import sys

import numpy as np


def sequences():
    rng = np.random.default_rng(3)
    B = "ACGT"

    def rnd(n):
        return "".join(B[i] for i in rng.integers(0, 4, n))

    seqs = []
    for _ in range(300):
        n = int(rng.integers(3, 400)); s = list(rnd(n))
        r = rng.random()
        if r < 0.15:
            for i in rng.integers(0, n, max(1, n // 20)): s[i] = "N"
        elif r < 0.3:
            for i in rng.integers(0, n, max(1, n // 15)): s[i] = "RYKMSWBDHVN"[int(rng.integers(0, 11))]
        elif r < 0.4:
            s = [c.lower() if rng.random() < 0.3 else c for c in s]
        elif r < 0.5:
            s = [("U" if c == "T" and rng.random() < 0.5 else c) for c in s]
        elif r < 0.55:
            s = [("u" if c == "T" and rng.random() < 0.5 else c.lower()) for c in s]
        elif r < 0.6:
            for i in rng.integers(0, n, 2): s[i] = "-*X.x"[int(rng.integers(0, 5))]
        seqs.append("".join(s))
    orf = "ATG" + "".join(rnd(3) for _ in range(60))
    seqs += ["ATGTAA", "TAATAGTGA", "ATG", "AT", "A", "", "ATGATGATGTAA" + rnd(50), rnd(1) + orf + "TAG" + rnd(2), orf * 3, "TTA" + orf[::-1],
             "NNNATGAAANNNTAA" * 4]
    seqs += [rnd(150) for _ in range(400)]
    return seqs


if __name__ == "__main__":
    out = sys.argv[1]
    data = b""; idx = []; hd = b""; hidx = []
    for i, s in enumerate(sequences()):
        e = s.encode() + b"\n\0"; idx.append("%d\t%d\t%d\n" % (2 * i + 1, len(data), len(e))); data += e
        h = ("read_%d desc\n" % i).encode() + b"\0"; hidx.append("%d\t%d\t%d\n" % (2 * i + 1, len(hd), len(h))); hd += h
    open(out, "wb").write(data); open(out + ".index", "w").write("".join(idx)); open(out + ".dbtype", "wb").write((1).to_bytes(4, "little"))
    open(out + "_h", "wb").write(hd); open(out + "_h.index", "w").write("".join(hidx)); open(out + "_h.dbtype", "wb").write((12).to_bytes(4, "little"))
