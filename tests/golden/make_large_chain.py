#!/usr/bin/env python3
"""Makes tests/golden/large_chain.json: checksums of every product of the protein chain on a 5 M-read sample of the BASELINE.json
configs[2] community model (skewed coverage: log-normal abundances, sigma 1), computed ENTIRELY by the CPU oracle:

    plass_oracle synthreads  (the read model of include/plasship_synth.h run on the CPU)
    -> extractorfs x2 -> translatenucs --add-orf-stop x2 -> concatdbs          (data/assemble.sh:41-77)
    -> ITERS x (kmermatcher -> rescorediagonal -> assembleresults)                (data/assemble.sh:110-175, without findassemblystart)

tests/test_gpu_large.py regenerates the reads on the GPU, runs the HIP path at the same size and compares the entry digest
(`plass_oracle dbsum`: sum over the entries of a 64-bit hash of key, length and bytes) of the read DB, the fragment DB and of
pref / aln / seq_{i+1} of every iteration.  Run here (no GPU needed): ~15 min on 8 cores, ~25 GB of /tmp.

    python tests/golden/make_large_chain.py [--pairs 2500000] [--iters 3]
"""
import argparse, hashlib, json, os, subprocess, sys, tempfile, time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def md5(path):
    h = hashlib.md5()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 24), b""):
            h.update(blk)
    return h.hexdigest()


def db_sums(path):
    """order-independent digest of the entries (plass_oracle dbsum) + md5 of the data / index files as the oracle's writer lays them out"""
    import __graft_entry__ as g
    out = subprocess.run([g.oracle_bin(), "dbsum", path], stdout=subprocess.PIPE, check=True, text=True).stdout.strip().split("\t")
    f = dict(x.split("=") for x in out[1:])
    return {"entries": int(f["entries"]), "bytes": int(f["bytes"]), "digest": f["digest"], "data_md5": md5(path), "index_md5": md5(path + ".index")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=2500000)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--threads", type=int, default=len(os.sched_getaffinity(0)))
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "large_chain.json"))
    a = ap.parse_args()
    import bench, __graft_entry__ as g
    from plass_amd import _lib
    subprocess.check_call(["make", "-j", "8"], cwd=os.path.join(ROOT, "oracle"))
    sp = bench.synth_params("c3", a.pairs)
    thr = ["--threads", str(a.threads)]
    res = {"made_by": "tests/golden/make_large_chain.py (CPU oracle only)", "config": "c3", "pairs": a.pairs, "iters": a.iters,
           "synth": {"n_pairs": sp.n_pairs, "seed": sp.seed, "n_genomes": sp.n_genomes, "genome_min_len": sp.genome_min_len, "genome_max_len": sp.genome_max_len,
                     "abundance_sigma": sp.abundance_sigma, "insert_mean": sp.insert_mean, "insert_sd": sp.insert_sd, "insert_min": sp.insert_min,
                     "read_len": sp.read_len, "error_rate": sp.error_rate}}
    with tempfile.TemporaryDirectory(dir=os.environ.get("TMPDIR", "/tmp")) as td:
        P = lambda n: os.path.join(td, n)
        t0 = time.time()
        print(g.run_oracle(["synthreads", P("reads"), "--pairs", str(sp.n_pairs), "--seed", str(sp.seed), "--genomes", str(sp.n_genomes),
                            "--genome-min-len", str(sp.genome_min_len), "--genome-max-len", str(sp.genome_max_len), "--abundance-sigma", repr(sp.abundance_sigma),
                            "--insert-mean", repr(sp.insert_mean), "--insert-sd", repr(sp.insert_sd), "--insert-min", str(sp.insert_min),
                            "--read-len", str(sp.read_len), "--error-rate", repr(sp.error_rate)]).strip())
        res["reads"] = db_sums(P("reads"))
        for name, par in (("long", _lib.PLASS_ORFS_LONG), ("start", _lib.PLASS_ORFS_START)):
            fl = []
            for k, v in par.items():
                fl += ["--" + k.replace("_", "-"), str(v)]
            print(g.run_oracle(["extractorfs", P("reads"), P("nucl_" + name)] + fl).strip())
            print(g.run_oracle(["translatenucs", P("nucl_" + name), P("aa_" + name), "--add-orf-stop", "1"]).strip())
            os.remove(P("nucl_" + name))
        print(g.run_oracle(["concatdbs", P("aa_long"), P("aa_start"), P("seq_0")]).strip())
        os.remove(P("reads")); os.remove(P("aa_long")); os.remove(P("aa_start"))
        res["fragments"] = db_sums(P("seq_0"))
        print("fragments:", res["fragments"], "%.0f s" % (time.time() - t0), flush=True)
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
            os.remove(s); os.remove(p); os.remove(al)
    with open(a.out, "w") as f:
        json.dump(res, f, indent=1)
        f.write("\n")
    print("wrote", a.out)


if __name__ == "__main__":
    main()
