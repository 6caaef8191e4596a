#!/usr/bin/env python3
"""GENERATION-TIME ONLY (build container).  How the reference resolves the nucleotide STRAND TIES of kmermatcher's sort #2, measured.

Sort #2 (kmermatcher.h:98-130) compares (rep, target, diagonal) only; nucleotide records of one such triple can differ in the strand
bit, and writeKmerMatcherResult (kmermatcher.cpp:866-893) reports the strand of the LAST record of the best diagonal's run — so the
`pref` line of a pair depends on where ips4o leaves tied records.  This script runs, on a fresh synthetic nucleotide read set
(deep, skewed coverage: that is where a pair shares k-mers on both strands),

    the UNMODIFIED reference  `penguin kmermatcher --threads 1`  and  `--threads 8`   (REF_BUILD, default /tmp/plass-build)
    the oracle with today's rule (ties in sort-#1 = k-mer order, oracle/kmermatcher.cpp) and with rounds 1-3's rule (reverse first)

for ITERS iterations of the nucleotide chain (each step fed the reference's previous output), and reports per iteration the
differing `pref` entries between the variants and the alignment lines `rescorediagonal` makes of each variant's `pref`.
It also cuts the small fixture tests/golden/strand_ties.tar.gz: the queries whose entries depend on the rule, with all their
targets, as a DB of their own, and what the reference (1 and 8 threads, identical there or the fixture is refused) writes for it.

    python tests/golden/make_strand_ties.py [--pairs 500000] [--iters 4] [--report profiles/r04_strand_ties.txt]
"""
import argparse, os, shutil, subprocess, sys, tarfile, tempfile, time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import dbdiff, dbcanon  # noqa: E402

KM = "--alph-size 5 --kmer-per-seq 60 --kmer-per-seq-scale 0.100 -k 22 -c 0 --cov-mode 0 --ignore-multi-kmer 1 --max-seq-len 200000 --hash-shift 67 --include-only-extendable 1".split()
RS = "--rescore-mode 3 -e 1e-05 -c 0 -a 0 --cov-mode 0 --min-seq-id 0.99 --min-aln-len 0 --seq-id-mode 0 --sort-results 0".split()
AS = "--min-seq-id 0.99 --max-seq-len 200000 --keep-target 1 --rescore-mode 3".split()


def run(cmd, log=None):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode:
        sys.stderr.write(r.stdout)
        raise SystemExit("failed: " + " ".join(cmd))
    if log is not None:
        log.append(r.stdout)
    return r.stdout


def differing(a, b):
    ta, ea = dbdiff.read_db(a)
    tb, eb = dbdiff.read_db(b)
    assert ta == tb and set(ea) == set(eb), (a, b)
    return sorted(k for k in ea if ea[k] != eb[k])


def differing_lines(a, b):
    """prefilter lines (query, target) whose text differs between two pref DBs, or that only one of them has"""
    _, ea = dbdiff.read_db(a)
    _, eb = dbdiff.read_db(b)
    n = 0
    for k in ea:
        if ea[k] != eb[k]:
            la = dict((l.split(b"\t", 1)[0], l) for l in ea[k][0].split(b"\n") if l.strip(b"\0"))
            lb = dict((l.split(b"\t", 1)[0], l) for l in eb[k][0].split(b"\n") if l.strip(b"\0"))
            n += sum(1 for t in set(la) | set(lb) if la.get(t) != lb.get(t))
    return n


def n_lines(db):
    _, e = dbdiff.read_db(db)
    return sum(v[0].count(b"\n") for v in e.values())


def rm(*paths):
    for p in paths:
        for f in os.listdir(os.path.dirname(p)):
            if f == os.path.basename(p) or f.startswith(os.path.basename(p) + "."):
                os.remove(os.path.join(os.path.dirname(p), f))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=500000)
    ap.add_argument("--iters", type=int, default=4)
    ap.add_argument("--seed", type=int, default=40427)
    ap.add_argument("--genomes", type=int, default=6)
    ap.add_argument("--sigma", type=float, default=1.3)
    ap.add_argument("--hairpins", type=int, default=0, help="plant N inverted repeats (arm 40-150 nt, loop 0-60 nt) in ONE gene-dense genome sized for --coverage "
                    "(numpy generator of plass_amd/synth.py) instead of the plass_oracle synthreads community: loci where reads share k-mers on both strands")
    ap.add_argument("--coverage", type=float, default=60.0)
    ap.add_argument("--report", default=os.path.join(ROOT, "profiles", "r04_strand_ties.txt"))
    ap.add_argument("--fixture", default=os.path.join(ROOT, "tests", "golden", "strand_ties.tar.gz"))
    a = ap.parse_args()
    ref = os.path.join(os.environ.get("REF_BUILD", "/tmp/plass-build"), "src", "penguin")
    subprocess.check_call(["make", "-j", "8"], cwd=os.path.join(ROOT, "oracle"), stdout=subprocess.DEVNULL)
    orc = os.path.join(ROOT, "oracle", "build", "plass_oracle")
    rep = []

    def say(s):
        print(s, flush=True)
        rep.append(s)

    say("# strand ties of kmermatcher's sort #2: unmodified reference (penguin, %s) vs oracle — tests/golden/make_strand_ties.py" % ref)
    if not a.hairpins:
        say("# reads: %d pairs 2x150 nt, %d genomes of 0.3-1.5 Mbp, log-normal abundances sigma %.1f, seed %d (plass_oracle synthreads)" % (a.pairs, a.genomes, a.sigma, a.seed))
    say("# kmermatcher %s" % " ".join(KM))
    fixture_done = False
    with tempfile.TemporaryDirectory(dir=os.environ.get("TMPDIR", "/tmp")) as td:
        P = lambda n: os.path.join(td, n)
        if a.hairpins:
            import numpy as np
            from plass_amd import synth
            reads, glen = synth.nucleotide_hairpin_reads(a.pairs, a.hairpins, a.seed, a.coverage)
            data, off, elen, key = synth.fixed_length_db(reads)
            synth.write_db(P("seq_0"), data, off, elen, key, 1)
            say("# reads: %d pairs 2x150 nt from ONE gene-dense genome of %d nt with %d planted inverted repeats (arm 40-150 nt, loop 0-60 nt), %.0fx coverage, 0.2 %% substitutions, seed %d (--hairpins; plass_amd/synth.py)"
                % (a.pairs, glen, a.hairpins, a.coverage, a.seed))
        else:
            run([orc, "synthreads", P("seq_0"), "--pairs", str(a.pairs), "--seed", str(a.seed), "--genomes", str(a.genomes), "--genome-min-len", "300000",
                 "--genome-max-len", "1500000", "--abundance-sigma", repr(a.sigma)])
        for it in range(a.iters):
            src = P("seq_%d" % it)
            t0 = time.time()
            run([ref, "kmermatcher", src, P("p_ref8")] + KM + ["--threads", "8", "-v", "1"])
            run([ref, "kmermatcher", src, P("p_ref1")] + KM + ["--threads", "1", "-v", "1"])
            log = []
            run([orc, "kmermatcher", src, P("p_new")] + KM + ["--threads", "8"], log)
            run([orc, "kmermatcher", src, P("p_old")] + KM + ["--threads", "8", "--oracle-old-strand-ties", "1"])
            ties = [l for l in log[0].splitlines() if "hold both strands" in l]
            d = {k: differing(P(x), P(y)) for k, (x, y) in {"ref1~ref8": ("p_ref1", "p_ref8"), "new~ref1": ("p_new", "p_ref1"), "new~ref8": ("p_new", "p_ref8"),
                                                               "old~ref1": ("p_old", "p_ref1"), "old~ref8": ("p_old", "p_ref8")}.items()}
            nent = len(dbdiff.read_db(P("p_ref8"))[1])
            alns = {}
            for v in ("ref8", "ref1", "new", "old"):
                run([orc, "rescorediagonal", src, src, P("p_" + v), P("a_" + v)] + RS + ["--threads", "8"])
                alns[v] = n_lines(P("a_" + v))
            say("iteration %d: %d entries; %s" % (it, nent, ties[0].split(": ", 1)[1] if ties else "no (rep, target, diagonal) triple holds both strands"))
            say("  differing pref entries: reference 1 thread vs 8 threads %d | oracle (k-mer order) vs reference %d / %d | oracle (reverse first, rounds 1-3) vs reference %d / %d"
                % (len(d["ref1~ref8"]), len(d["new~ref1"]), len(d["new~ref8"]), len(d["old~ref1"]), len(d["old~ref8"])))
            dl = {k: differing_lines(P(x), P(y)) for k, (x, y) in {"ref1~ref8": ("p_ref1", "p_ref8"), "new~ref1": ("p_new", "p_ref1"), "new~ref8": ("p_new", "p_ref8"),
                                                                     "old~ref1": ("p_old", "p_ref1"), "old~ref8": ("p_old", "p_ref8")}.items()}
            say("  differing pref LINES (query, target):  reference 1 thread vs 8 threads %d | oracle (k-mer order) vs reference %d / %d | oracle (reverse first, rounds 1-3) vs reference %d / %d"
                % (dl["ref1~ref8"], dl["new~ref1"], dl["new~ref8"], dl["old~ref1"], dl["old~ref8"]))
            say("  alignment lines rescorediagonal writes from each pref: reference-8 %d, reference-1 %d, oracle (k-mer order) %d, oracle (reverse first) %d"
                % (alns["ref8"], alns["ref1"], alns["new"], alns["old"]))
            # ---- the fixture: queries whose entry depends on the rule, with their targets, as a DB of their own ----
            dep = sorted(set(d["old~ref8"]) | set(d["old~ref1"]))
            pe = se = None
            for qi, q0 in enumerate(dep[:12] if not fixture_done else []):
                if fixture_done:
                    break
                if pe is None:
                    _, pe = dbdiff.read_db(P("p_ref8"))
                    _, se = dbdiff.read_db(src)
                keys = set()
                for q in [q0]:
                    keys.add(q)
                    for ln in pe[q][0].split(b"\n"):
                        if ln.strip(b"\0"):
                            keys.add(int(ln.split(b"\t")[0]))
                dep = [q0]
                fd = os.path.join(td, "fx", "strand_ties")
                os.makedirs(fd, exist_ok=True)
                off = 0
                with open(os.path.join(fd, "seq_0"), "wb") as f, open(os.path.join(fd, "seq_0.index"), "wb") as fi:
                    for k in sorted(keys):
                        f.write(se[k][0]); fi.write(b"%d\t%d\t%d\n" % (k, off, len(se[k][0]))); off += len(se[k][0])
                shutil.copy(src + ".dbtype", os.path.join(fd, "seq_0.dbtype"))
                s0 = os.path.join(fd, "seq_0")
                run([ref, "kmermatcher", s0, P("fx_p8")] + KM + ["--threads", "8", "-v", "1"])
                run([ref, "kmermatcher", s0, P("fx_p1")] + KM + ["--threads", "1", "-v", "1"])
                flog = []
                run([orc, "kmermatcher", s0, P("fx_pn")] + KM, flog)
                run([orc, "kmermatcher", s0, P("fx_po")] + KM + ["--oracle-old-strand-ties", "1"])
                r18, rn, ro = differing(P("fx_p1"), P("fx_p8")), differing(P("fx_pn"), P("fx_p8")), differing(P("fx_po"), P("fx_p8"))
                say("  fixture candidate (%d sequences around queries %s): reference 1 vs 8 threads %d differing, oracle (k-mer order) vs reference %d, oracle (reverse first) vs reference %d; %s"
                    % (len(keys), dep[:6], len(r18), len(rn), len(ro), [l for l in flog[0].splitlines() if "hold both strands" in l][:1]))
                if not r18 and not rn and ro:
                    dbcanon.canon(P("fx_p8"), os.path.join(fd, "pref_0"))
                    run([ref, "rescorediagonal", s0, s0, P("fx_p8"), P("fx_a")] + RS + ["--threads", "4", "-v", "1"])
                    dbcanon.canon(P("fx_a"), os.path.join(fd, "aln_0"))
                    with open(os.path.join(fd, "MANIFEST"), "w") as f:
                        f.write("nucleotide strand ties (kmermatcher.h:98-130, kmermatcher.cpp:866-893): %d sequences of a synthetic read set (make_strand_ties.py, seed %d,\n"
                                "iteration %d) around queries whose (rep, target, diagonal) runs hold records of both strands.\n"
                                "seq_0 -> pref_0 [penguin kmermatcher %s] -> aln_0 [rescorediagonal %s]\n"
                                "written by the UNMODIFIED reference; identical at --threads 1 and --threads 8.  Rounds 1-3's tie rule (reverse strand first)\n"
                                "gets %d of the pref entries wrong on this DB.\n" % (len(keys), a.seed, it, " ".join(KM), " ".join(RS), len(ro)))
                    with tarfile.open(a.fixture, "w:gz") as tf:
                        tf.add(fd, arcname="strand_ties")
                    say("  -> wrote %s" % os.path.relpath(a.fixture, ROOT))
                    fixture_done = True
            # ---- next input: the reference's own chain ----
            if it + 1 < a.iters:
                run([ref, "rescorediagonal", src, src, P("p_ref8"), P("a_chain")] + RS + ["--threads", "8", "-v", "1"])
                run([ref, "nuclassembleresults", src, P("a_chain"), P("s_next")] + AS + ["--threads", "8", "-v", "1"])
                dbcanon.canon(P("s_next"), P("seq_%d" % (it + 1)))
                rm(P("a_chain"), P("s_next"))
            for v in ("ref8", "ref1", "new", "old"):
                rm(P("p_" + v), P("a_" + v))
            say("  (%.0f s)" % (time.time() - t0))
    os.makedirs(os.path.dirname(a.report), exist_ok=True)
    with open(a.report, "w") as f:
        f.write("\n".join(rep) + "\n")
    print("wrote", a.report)


if __name__ == "__main__":
    main()
