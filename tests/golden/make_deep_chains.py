#!/usr/bin/env python3
"""Makes tests/golden/deep_chains.json: digests of every product of DEEP chains — the regime the headline number is measured in
 — computed ENTIRELY by the CPU oracle on reads `plass_oracle synthreads` generates:

  c2_exact  BASELINE.json configs[1] as stated: 1 M reads (500 000 pairs, seed 1), the six iterations of `plass assemble
            --num-iterations 6` INCLUDING iteration 0's findassemblystart pass (data/assemble.sh:85-156: kmermatcher, rescorediagonal,
            findassemblystart, kmermatcher, rescorediagonal, assembleresults), hash shifts of src/workflow/Assembler.cpp:99-110
  c2_bench  the same reads through the chain `bench.py --config c2` times (no findassemblystart): tests/golden/c2_chain_digests.json, what that
            bench line's `verify` compares its run with
  c3_deep   2 M reads (1 M pairs) of the configs[2] community model (skewed coverage), the TWELVE iterations of the default
            `plass assemble` chain as bench.py runs it (hash shifts 67, 68, 68, 69, ...; contigs of thousands of residues, queues of
            more than 64 alignments, the selected-window cache alternating with re-seeded iterations, the DB heap alternating between
            append and full copy)
  c5_deep   2 M reads (1 M pairs) of the configs[4] model: SIX iterations of the nucleotide chain (data/nuclassemble.sh:95-137:
            kmermatcher -k 22, rescorediagonal, nuclassembleresults, cyclecheck --chop-cycle 1, the non-circular rest) and FOUR of the
            protein-guided chain (data/guidedNuclAssemble.sh:44-126)

tests/test_gpu_deep.py regenerates the reads on the GPU, runs the HIP path and compares the `plass_oracle dbsum` digest of every DB.
Run here (no GPU needed; about an hour on 8 cores):

    python tests/golden/make_deep_chains.py [--only c2_exact,c3_deep,c5_deep]
"""
import argparse, json, os, subprocess, sys, tempfile, time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

from make_large_nucl import db_sums, rest_db, rm          # noqa: E402


def last(e):
    return e.strip().splitlines()[-1]


def synth(g, sp, path):
    return g.run_oracle(["synthreads", path, "--pairs", str(sp.n_pairs), "--seed", str(sp.seed), "--genomes", str(sp.n_genomes),
                         "--genome-min-len", str(sp.genome_min_len), "--genome-max-len", str(sp.genome_max_len), "--abundance-sigma", repr(sp.abundance_sigma),
                         "--insert-mean", repr(sp.insert_mean), "--insert-sd", repr(sp.insert_sd), "--insert-min", str(sp.insert_min),
                         "--read-len", str(sp.read_len), "--error-rate", repr(sp.error_rate)]).strip()


def synth_dict(sp):
    return {"n_pairs": sp.n_pairs, "seed": sp.seed, "n_genomes": sp.n_genomes, "genome_min_len": sp.genome_min_len, "genome_max_len": sp.genome_max_len,
            "abundance_sigma": sp.abundance_sigma, "insert_mean": sp.insert_mean, "insert_sd": sp.insert_sd, "insert_min": sp.insert_min,
            "read_len": sp.read_len, "error_rate": sp.error_rate}


def km_flags(bench, it):
    return ["--alph-size", "13", "--kmer-per-seq", "60", "--kmer-per-seq-scale", "0", "-k", "14", "-c", "0", "--hash-shift", str(bench.hash_shift(it)),
            "--include-only-extendable", "1" if it else "0", "--ignore-multi-kmer", "1"]


RS = ["--rescore-mode", "3", "--min-seq-id", "0.9", "-e", "1e-5", "-c", "0"]
AS = ["--min-seq-id", "0.9", "--max-seq-len", "65535", "--keep-target", "1"]


def protein_chain(g, bench, _lib, cfg, pairs, iters, findstart, thr, td, t0):
    P = lambda n: os.path.join(td, n)
    sp = bench.synth_params(cfg, pairs)
    res = {"config": cfg, "pairs": pairs, "iters": iters, "findassemblystart": findstart, "synth": synth_dict(sp)}
    print(synth(g, sp, P("reads")))
    res["reads"] = db_sums(P("reads"))
    for name, par in (("long", _lib.PLASS_ORFS_LONG), ("start", _lib.PLASS_ORFS_START)):
        fl = []
        for k, v in par.items():
            fl += ["--" + k.replace("_", "-"), str(v)]
        g.run_oracle(["extractorfs", P("reads"), P("nucl_" + name)] + fl)
        g.run_oracle(["translatenucs", P("nucl_" + name), P("aa_" + name), "--add-orf-stop", "1"])
        rm(P("nucl_" + name), P("nucl_" + name + "_h"))
    g.run_oracle(["concatdbs", P("aa_long"), P("aa_start"), P("seq_0")])
    rm(P("reads"), P("aa_long"), P("aa_start"))
    res["fragments"] = db_sums(P("seq_0"))
    print(cfg, "fragments:", res["fragments"], "%.0f s" % (time.time() - t0), flush=True)
    res["iterations"] = []
    for it in range(iters):
        s, p, al, o = P("seq_%d" % it), P("pref"), P("aln"), P("seq_%d" % (it + 1))
        km = km_flags(bench, it)
        row = {}
        e = [g.run_oracle(["kmermatcher", s, p] + km + thr), g.run_oracle(["rescorediagonal", s, s, p, al] + RS + thr)]
        if it == 0 and findstart:
            row["pref_uncorrected"] = db_sums(p); row["aln_uncorrected"] = db_sums(al)
            e.append(g.run_oracle(["findassemblystart", s, al, P("corrected")] + thr))
            row["corrected"] = db_sums(P("corrected"))
            rm(s, p, al)
            s = P("corrected")
            e += [g.run_oracle(["kmermatcher", s, p] + km + thr), g.run_oracle(["rescorediagonal", s, s, p, al] + RS + thr)]
        e.append(g.run_oracle(["assembleresults", s, al, o] + AS + thr))
        row.update({"pref": db_sums(p), "aln": db_sums(al), "seq": db_sums(o), "oracle": [last(x) for x in e]})
        res["iterations"].append(row)
        print(cfg, it, row, "%.0f s" % (time.time() - t0), flush=True)
        rm(s, p, al)
    rm(P("seq_%d" % iters))
    return res


def nucl_chains(g, bench, _lib, T, pairs, iters, giters, thr, td, t0):
    P = lambda n: os.path.join(td, n)
    sp = bench.synth_params("c5", pairs)
    res = {"config": "c5", "pairs": pairs, "synth": synth_dict(sp)}
    print(synth(g, sp, P("reads")))
    res["reads"] = db_sums(P("reads"))
    res["nucl"] = []
    src = P("reads")
    for it in range(iters):
        p, al, o, cy, rest = P("pref"), P("aln"), P("assembly_%d" % it), P("cycle_%d" % it), P("rest_%d" % it)
        e1 = g.run_oracle(["kmermatcher", src, p] + T.NUCL_KM + thr)
        e2 = g.run_oracle(["rescorediagonal", src, src, p, al] + T.NUCL_RS + thr)
        e3 = g.run_oracle(["nuclassembleresults", src, al, o] + T.NUCL_AS[:6] + thr)
        e4 = g.run_oracle(["cyclecheck", o, cy, "--max-seq-len", "200000", "--chop-cycle", "1"] + thr)
        ncyc = rest_db(o, cy, rest)
        row = {"pref": db_sums(p), "aln": db_sums(al), "assembly": db_sums(o), "cycle": db_sums(cy), "rest": db_sums(rest), "n_cyclic": ncyc,
               "oracle": [last(x) for x in (e1, e2, e3, e4)]}
        res["nucl"].append(row)
        print("nucl", it, row, "%.0f s" % (time.time() - t0), flush=True)
        rm(p, al, cy)
        if it:
            rm(P("rest_%d" % (it - 1)), P("assembly_%d" % (it - 1)))
        src = rest
    rm(src, P("assembly_%d" % (iters - 1)))
    for name, par in (("long", _lib.PLASS_ORFS_LONG), ("start", _lib.PLASS_ORFS_START)):
        fl = []
        for k, v in par.items():
            fl += ["--" + k.replace("_", "-"), str(v)]
        g.run_oracle(["extractorfs", P("reads"), P("nucl_" + name)] + fl)
    g.run_oracle(["concatdbs", P("nucl_long"), P("nucl_start"), P("nucl_0")])
    g.run_oracle(["concatdbs", P("nucl_long_h"), P("nucl_start_h"), P("nucl_0_h")])
    g.run_oracle(["translatenucs", P("nucl_0"), P("aa_0"), "--add-orf-stop", "1"])
    rm(P("reads"), P("nucl_long"), P("nucl_start"), P("nucl_long_h"), P("nucl_start_h"))
    res["guided_input"] = {"nucl": db_sums(P("nucl_0")), "aa": db_sums(P("aa_0"))}
    res["guided"] = []
    for it in range(giters):
        nu, aa, p, al, an = P("nucl_%d" % it), P("aa_%d" % it), P("pref"), P("aln"), P("aln_nucl")
        nu2, aa2 = P("nucl_%d" % (it + 1)), P("aa_%d" % (it + 1))
        e1 = g.run_oracle(["kmermatcher", aa, p] + T.GD_KM + thr)
        e2 = g.run_oracle(["rescorediagonal", aa, aa, p, al] + T.GD_RS + thr)
        e3 = g.run_oracle(["proteinaln2nucl", nu, nu, aa, aa, al, an] + T.GD_P2N + thr)
        e4 = g.run_oracle(["guidedassembleresults", nu, aa, an, nu2, aa2] + T.GD_AS[:6] + thr)
        row = {"pref": db_sums(p), "aln": db_sums(al), "aln_nucl": db_sums(an), "nucl": db_sums(nu2), "aa": db_sums(aa2),
               "oracle": [last(x) for x in (e1, e2, e3, e4)]}
        res["guided"].append(row)
        print("guided", it, row, "%.0f s" % (time.time() - t0), flush=True)
        rm(nu, aa, p, al, an)
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="c2_exact,c3_deep,c5_deep")
    ap.add_argument("--deep-pairs", type=int, default=1000000)
    ap.add_argument("--nucl-pairs", type=int, default=1000000)
    ap.add_argument("--threads", type=int, default=len(os.sched_getaffinity(0)))
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "deep_chains.json"))
    a = ap.parse_args()
    import bench, __graft_entry__ as g
    import conftest as T
    from plass_amd import _lib
    subprocess.check_call(["make", "-j", "8"], cwd=os.path.join(ROOT, "oracle"))
    thr = ["--threads", str(a.threads)]
    res = json.load(open(a.out)) if os.path.exists(a.out) else {}
    res["made_by"] = "tests/golden/make_deep_chains.py (CPU oracle only)"
    t0 = time.time()
    for what in a.only.split(","):
        with tempfile.TemporaryDirectory(dir=os.environ.get("TMPDIR", "/tmp")) as td:
            if what == "c2_exact":
                res[what] = protein_chain(g, bench, _lib, "c2", bench.CONFIGS["c2"][1], 6, True, thr, td, t0)
            elif what == "c2_bench":        # the chain `bench.py --config c2` times (no findassemblystart): its `verify` digests, by the oracle
                res[what] = protein_chain(g, bench, _lib, "c2", bench.CONFIGS["c2"][1], 6, False, thr, td, t0)
                with open(os.path.join(ROOT, "tests", "golden", "c2_chain_digests.json"), "w") as f:
                    json.dump({"made_by": "tests/golden/make_deep_chains.py --only c2_bench (CPU oracle: the six iterations of bench.py --config c2, digests of seq_1 .. seq_6)",
                               "config": "c2", "pairs": res[what]["pairs"], "digests": [r["seq"]["digest"] for r in res[what]["iterations"]]}, f, indent=1)
                    f.write("\n")
            elif what == "c3_deep":
                res[what] = protein_chain(g, bench, _lib, "c3", a.deep_pairs, 12, False, thr, td, t0)
            elif what == "c5_deep":
                res[what] = nucl_chains(g, bench, _lib, T, a.nucl_pairs, 6, 4, thr, td, t0)
            else:
                raise SystemExit("unknown section " + what)
        with open(a.out, "w") as f:
            json.dump(res, f, indent=1)
            f.write("\n")
        print("wrote", what, "->", a.out, "%.0f s" % (time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
