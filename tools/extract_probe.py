#!/usr/bin/env python3
"""Development probe: extraction time of kmermatcher as a function of sequence length (random protein sequences of one length:
t(L) = per-sequence overhead + per-window cost).  Usage: tools/extract_probe.py [residues in total, default 4e8]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import plass_amd

AA = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", dtype=np.uint8)


def make_db(n, L, seed):
    rng = np.random.default_rng(seed)
    seq = AA[rng.integers(0, 20, size=(n, L), dtype=np.uint8)]
    ent = np.zeros((n, L + 2), dtype=np.uint8); ent[:, :L] = seq; ent[:, L] = 10
    off = np.arange(n, dtype=np.uint64) * np.uint64(L + 2)
    return ent.tobytes(), off, np.full(n, L + 2, dtype=np.uint32), np.arange(n, dtype=np.uint32)


def main():
    total = int(float(sys.argv[1])) if len(sys.argv) > 1 else 400_000_000
    ctx = plass_amd.Context(0)
    print("%6s %9s | %10s %10s | %12s %12s | %s" % ("L", "seqs", "short ms", "wave ms", "ns/seq", "ps/window", "stage ms: part group sort reduce"))
    lens = [int(x) for x in os.environ.get("PROBE_LENGTHS", "48,60,80,100,150,250,400,700,1000,1500,2500").split(",")]
    for L in lens:
        n = max(1000, total // L)
        data, off, elen, key = make_db(n, L, 7 + L)
        db = ctx.upload_seqdb(data, off, elen, key, 0)
        best = None
        for rep in range(3):
            c, st = ctx.kmermatcher(db, plass_amd.KmermatchParams(hash_shift=67, include_only_extendable=True))
            c.free()
            if best is None or st.ms_extract_short_kernel + st.ms_extract_wave_kernel < best[0] + best[1]:
                best = (st.ms_extract_short_kernel, st.ms_extract_wave_kernel, st.ms_sort1, st.ms_group, st.ms_sort2, st.ms_reduce)
        t = best[0] + best[1]
        print("%6d %9d | %10.3f %10.3f | %12.1f %12.1f | %.2f %.2f %.2f %.2f" % (L, n, best[0], best[1], t * 1e6 / n, t * 1e9 / (n * max(L - 13, 1)), best[2], best[3], best[4], best[5]), flush=True)
        db.free()
    ctx.close()


if __name__ == "__main__":
    main()
