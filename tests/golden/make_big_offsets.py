#!/usr/bin/env python3
"""Makes tests/golden/big_offsets.json: CPU-oracle digests of a protein chain on a DB whose DATA EXCEEDS 2^32 BYTES and whose live
sequences all lie above that mark (missing #3: the 64-bit offset paths of extraction, rescoring, the extension arena and the
DB writer had only ever been compared with themselves at sizes the oracle cannot follow).

    filler:   FILLERS sequences of FILLER_LEN random residues (uniform over the 20 amino acids, numpy PCG64, seed 9): nothing overlaps
              anything, every filler contributes its <= 59 lowest-hash k-mers + identity record and is carried through unchanged.
              Default 560 000 x 8 000 = 4.48 G residues, 4.4812 GB of entries — the first live sequence starts 186 MB above 2^32.
              (8 000 residues: the longest the LDS-resident extraction tier takes; all lengths stay below 32 767, so the run keeps the
              16-byte KmerPosition<short> records of the headline configuration.)
    live:     the protein fragments of PAIRS synthetic read pairs of the configs[2] community (plass_oracle synthreads -> extractorfs x2
              -> translatenucs --add-orf-stop -> concatdbs, as tests/golden/make_large_chain.py)
    DB:       concatdbs filler live   (the fragments' keys and offsets follow the fillers')
    chain:    ITERS x (kmermatcher -> rescorediagonal -> assembleresults), digests (plass_oracle dbsum) of pref / aln / seq_{i+1}

tests/test_gpu_large.py::test_offsets_beyond_4gib_against_oracle_checksums builds the same DB on the GPU box (the fillers with the same
numpy generator, the reads with the GPU generator) and compares every DB.  Run here (no GPU needed): ~10 min on 8 cores, ~30 GB of /tmp.

    python tests/golden/make_big_offsets.py [--pairs 500000] [--iters 3]
"""
import argparse, json, os, subprocess, sys, tempfile, time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

FILLERS, FILLER_LEN, FILLER_SEED = 560000, 8000, 9


def write_filler_db(path, n=FILLERS, length=FILLER_LEN, seed=FILLER_SEED, chunk=20000):
    """n entries of `length` random residues, keys 0..n-1, written chunk by chunk (the whole DB is n * (length + 2) bytes)"""
    import numpy as np
    aa = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", dtype=np.uint8)
    rng = np.random.default_rng(seed)
    with open(path, "wb") as f:
        for c0 in range(0, n, chunk):
            m = min(chunk, n - c0)
            ent = np.empty((m, length + 2), dtype=np.uint8)
            ent[:, :length] = aa[rng.integers(0, 20, size=(m, length), dtype=np.uint8)]
            ent[:, length] = 10; ent[:, length + 1] = 0
            f.write(ent.tobytes())
    with open(path + ".index", "wb") as f:
        e = length + 2
        f.write(b"".join(b"%d\t%d\t%d\n" % (i, i * e, e) for i in range(n)))
    with open(path + ".dbtype", "wb") as f:
        f.write((0).to_bytes(4, "little"))
    return n * (length + 2)


def db_sums(path):
    import __graft_entry__ as g
    out = subprocess.run([g.oracle_bin(), "dbsum", path], stdout=subprocess.PIPE, check=True, text=True).stdout.strip().split("\t")
    f = dict(x.split("=") for x in out[1:])
    return {"entries": int(f["entries"]), "bytes": int(f["bytes"]), "digest": f["digest"]}


def rm(*paths):
    for p in paths:
        for sfx in ("", ".index", ".dbtype"):
            if os.path.lexists(p + sfx):
                os.remove(p + sfx)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=500000)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--threads", type=int, default=len(os.sched_getaffinity(0)))
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "big_offsets.json"))
    a = ap.parse_args()
    import bench, __graft_entry__ as g
    from plass_amd import _lib
    subprocess.check_call(["make", "-j", "8"], cwd=os.path.join(ROOT, "oracle"), stdout=subprocess.DEVNULL)
    sp = bench.synth_params("c3", a.pairs)
    thr = ["--threads", str(a.threads)]
    res = {"made_by": "tests/golden/make_big_offsets.py (CPU oracle only)", "config": "c3", "pairs": a.pairs, "iters": a.iters,
           "filler": {"n": FILLERS, "length": FILLER_LEN, "seed": FILLER_SEED},
           "synth": {"n_pairs": sp.n_pairs, "seed": sp.seed, "n_genomes": sp.n_genomes, "genome_min_len": sp.genome_min_len, "genome_max_len": sp.genome_max_len,
                     "abundance_sigma": sp.abundance_sigma, "insert_mean": sp.insert_mean, "insert_sd": sp.insert_sd, "insert_min": sp.insert_min,
                     "read_len": sp.read_len, "error_rate": sp.error_rate}}
    with tempfile.TemporaryDirectory(dir=os.environ.get("TMPDIR", "/tmp")) as td:
        P = lambda n: os.path.join(td, n)
        t0 = time.time()
        fb = write_filler_db(P("filler"))
        assert fb > (1 << 32), "the fillers must push the live sequences beyond 2^32 bytes"
        res["filler"]["bytes"] = fb
        res["filler_db"] = db_sums(P("filler"))
        print("filler:", res["filler_db"], "%.0f s" % (time.time() - t0), flush=True)
        print(g.run_oracle(["synthreads", P("reads"), "--pairs", str(sp.n_pairs), "--seed", str(sp.seed), "--genomes", str(sp.n_genomes),
                            "--genome-min-len", str(sp.genome_min_len), "--genome-max-len", str(sp.genome_max_len), "--abundance-sigma", repr(sp.abundance_sigma),
                            "--insert-mean", repr(sp.insert_mean), "--insert-sd", repr(sp.insert_sd), "--insert-min", str(sp.insert_min),
                            "--read-len", str(sp.read_len), "--error-rate", repr(sp.error_rate)]).strip())
        for name, par in (("long", _lib.PLASS_ORFS_LONG), ("start", _lib.PLASS_ORFS_START)):
            fl = []
            for k, v in par.items():
                fl += ["--" + k.replace("_", "-"), str(v)]
            g.run_oracle(["extractorfs", P("reads"), P("nucl_" + name)] + fl)
            g.run_oracle(["translatenucs", P("nucl_" + name), P("aa_" + name), "--add-orf-stop", "1"])
            rm(P("nucl_" + name), P("nucl_" + name + "_h"))
        g.run_oracle(["concatdbs", P("aa_long"), P("aa_start"), P("live")])
        rm(P("reads"), P("aa_long"), P("aa_start"))
        res["live"] = db_sums(P("live"))
        g.run_oracle(["concatdbs", P("filler"), P("live"), P("seq_0")])
        rm(P("filler"), P("live"))
        res["db"] = db_sums(P("seq_0"))
        print("db:", res["db"], "live:", res["live"], "%.0f s" % (time.time() - t0), flush=True)
        res["iterations"] = []
        for it in range(a.iters):
            s, p, al, o = P("seq_%d" % it), P("pref"), P("aln"), P("seq_%d" % (it + 1))
            km = ["--alph-size", "13", "--kmer-per-seq", "60", "--kmer-per-seq-scale", "0", "-k", "14", "-c", "0", "--hash-shift", str(bench.hash_shift(it)),
                  "--include-only-extendable", "1" if it else "0", "--ignore-multi-kmer", "1"]
            e1 = g.run_oracle(["kmermatcher", s, p] + km + thr)
            e2 = g.run_oracle(["rescorediagonal", s, s, p, al, "--rescore-mode", "3", "--min-seq-id", "0.9", "-e", "1e-5", "-c", "0"] + thr)
            e3 = g.run_oracle(["assembleresults", s, al, o, "--min-seq-id", "0.9", "--max-seq-len", "65535", "--keep-target", "1"] + thr)
            row = {"pref": db_sums(p), "aln": db_sums(al), "seq": db_sums(o), "oracle": [e1.strip().splitlines()[-1], e2.strip().splitlines()[-1], e3.strip().splitlines()[-1]]}
            res["iterations"].append(row)
            print(it, row, "%.0f s" % (time.time() - t0), flush=True)
            rm(s, p, al)
    with open(a.out, "w") as f:
        json.dump(res, f, indent=1)
        f.write("\n")
    print("wrote", a.out)


if __name__ == "__main__":
    main()
