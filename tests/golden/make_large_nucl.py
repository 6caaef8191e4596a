#!/usr/bin/env python3
"""Makes tests/golden/large_nucl.json: checksums of every product of PenguiN's two nucleotide-level chains (BASELINE.json configs[4])
on a 2 M-read sample of the community model (log-normal abundances, sigma 1, seed 3), computed ENTIRELY by the CPU oracle:

    plass_oracle synthreads                                             (the read model of include/plasship_synth.h on the CPU)
    nucleotide chain (data/nuclassemble.sh:99-137), ITERS x:
        kmermatcher -k 22 -> rescorediagonal -> nuclassembleresults -> cyclecheck --chop-cycle 1 -> the non-circular rest
    protein-guided chain (data/guidedNuclAssemble.sh:44-124):
        extractorfs x2 -> concatdbs (sequences and headers) -> translatenucs --add-orf-stop, then GITERS x:
        kmermatcher -k 14 -> rescorediagonal -a 1 -> proteinaln2nucl -> guidedassembleresults

tests/test_gpu_large_nucl.py regenerates the reads on the GPU, runs the HIP path at the same size and compares the entry digest
(`plass_oracle dbsum`) of every DB.  Run here (no GPU needed):

    python tests/golden/make_large_nucl.py [--pairs 1000000] [--iters 3] [--guided-iters 2]
"""
import argparse, json, os, subprocess, sys, tempfile, time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def db_sums(path):
    import __graft_entry__ as g
    out = subprocess.run([g.oracle_bin(), "dbsum", path], stdout=subprocess.PIPE, check=True, text=True).stdout.strip().split("\t")
    f = dict(x.split("=") for x in out[1:])
    return {"entries": int(f["entries"]), "bytes": int(f["bytes"]), "digest": f["digest"]}


def rest_db(assembly, cycle, out):
    """what data/nuclassemble.sh:24-31 does with awk: the entries of `assembly` whose key has no entry in `cycle` (same data file)"""
    cyc = set(int(l.split(b"\t", 1)[0]) for l in open(cycle + ".index", "rb"))
    with open(out + ".index", "wb") as f:
        for l in open(assembly + ".index", "rb"):
            if int(l.split(b"\t", 1)[0]) not in cyc:
                f.write(l)
    for sfx in ("", ".dbtype"):
        if os.path.lexists(out + sfx):
            os.remove(out + sfx)
        os.symlink(assembly + sfx, out + sfx)
    return len(cyc)


def rm(*paths):
    for p in paths:
        for sfx in ("", ".index", ".dbtype"):
            if os.path.lexists(p + sfx):
                os.remove(p + sfx)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=1000000)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--guided-iters", type=int, default=2)
    ap.add_argument("--threads", type=int, default=len(os.sched_getaffinity(0)))
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "large_nucl.json"))
    a = ap.parse_args()
    import bench, __graft_entry__ as g
    import conftest as T
    from plass_amd import _lib
    subprocess.check_call(["make", "-j", "8"], cwd=os.path.join(ROOT, "oracle"))
    sp = bench.synth_params("c5", a.pairs)
    thr = ["--threads", str(a.threads)]
    res = {"made_by": "tests/golden/make_large_nucl.py (CPU oracle only)", "config": "c5", "pairs": a.pairs,
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
        # ---- nucleotide chain ----
        res["nucl"] = []
        src = P("reads")
        for it in range(a.iters):
            p, al, o, cy, rest = P("pref"), P("aln"), P("assembly_%d" % it), P("cycle_%d" % it), P("rest_%d" % it)
            e1 = g.run_oracle(["kmermatcher", src, p] + T.NUCL_KM + thr)
            e2 = g.run_oracle(["rescorediagonal", src, src, p, al] + T.NUCL_RS + thr)
            e3 = g.run_oracle(["nuclassembleresults", src, al, o] + T.NUCL_AS[:6] + thr)
            e4 = g.run_oracle(["cyclecheck", o, cy, "--max-seq-len", "200000", "--chop-cycle", "1"] + thr)
            ncyc = rest_db(o, cy, rest)
            row = {"pref": db_sums(p), "aln": db_sums(al), "assembly": db_sums(o), "cycle": db_sums(cy), "rest": db_sums(rest), "n_cyclic": ncyc,
                   "oracle": [e.strip().splitlines()[-1] for e in (e1, e2, e3, e4)]}
            res["nucl"].append(row)
            print("nucl", it, row, "%.0f s" % (time.time() - t0), flush=True)
            rm(p, al, cy)
            if it:
                rm(P("rest_%d" % (it - 1)), P("assembly_%d" % (it - 1)))
            src = rest
        rm(src, P("assembly_%d" % (a.iters - 1)))
        # ---- protein-guided chain ----
        for name, par in (("long", _lib.PLASS_ORFS_LONG), ("start", _lib.PLASS_ORFS_START)):
            fl = []
            for k, v in par.items():
                fl += ["--" + k.replace("_", "-"), str(v)]
            print(g.run_oracle(["extractorfs", P("reads"), P("nucl_" + name)] + fl).strip())
        print(g.run_oracle(["concatdbs", P("nucl_long"), P("nucl_start"), P("nucl_0")]).strip())
        print(g.run_oracle(["concatdbs", P("nucl_long_h"), P("nucl_start_h"), P("nucl_0_h")]).strip())
        print(g.run_oracle(["translatenucs", P("nucl_0"), P("aa_0"), "--add-orf-stop", "1"]).strip())
        rm(P("reads"), P("nucl_long"), P("nucl_start"), P("nucl_long_h"), P("nucl_start_h"))
        res["guided_input"] = {"nucl": db_sums(P("nucl_0")), "aa": db_sums(P("aa_0"))}
        print("guided input:", res["guided_input"], "%.0f s" % (time.time() - t0), flush=True)
        res["guided"] = []
        for it in range(a.guided_iters):
            nu, aa, p, al, an = P("nucl_%d" % it), P("aa_%d" % it), P("pref"), P("aln"), P("aln_nucl")
            nu2, aa2 = P("nucl_%d" % (it + 1)), P("aa_%d" % (it + 1))
            e1 = g.run_oracle(["kmermatcher", aa, p] + T.GD_KM + thr)
            e2 = g.run_oracle(["rescorediagonal", aa, aa, p, al] + T.GD_RS + thr)
            e3 = g.run_oracle(["proteinaln2nucl", nu, nu, aa, aa, al, an] + T.GD_P2N + thr)
            e4 = g.run_oracle(["guidedassembleresults", nu, aa, an, nu2, aa2] + T.GD_AS[:6] + thr)
            row = {"pref": db_sums(p), "aln": db_sums(al), "aln_nucl": db_sums(an), "nucl": db_sums(nu2), "aa": db_sums(aa2),
                   "oracle": [e.strip().splitlines()[-1] for e in (e1, e2, e3, e4)]}
            res["guided"].append(row)
            print("guided", it, row, "%.0f s" % (time.time() - t0), flush=True)
            rm(nu, aa, p, al, an)
    with open(a.out, "w") as f:
        json.dump(res, f, indent=1)
        f.write("\n")
    print("wrote", a.out)


if __name__ == "__main__":
    main()
