#!/usr/bin/env python3
"""BUILD CONTAINER ONLY (needs the unmodified reference, REF_BUILD = /tmp/plass-build).  Makes tests/golden/strand_membership.json.

VERDICT r5 item 5a.  The reference's NUCLEOTIDE kmermatcher is not deterministic on pairs whose (representative, target, diagonal)
triple holds k-mer records of both strands: its second sort compares (rep, target, diagonal) only (kmermatcher.h:98-130), ips4o is not
stable, and the strand a pair is reported with is that of whichever tied record ends up last (kmermatcher.cpp:866-893) — the judge's
seven runs of one command on one input gave seven versions of one entry, three of them at --threads 1.  "Bit-identical" therefore
cannot be asked of such an entry; what CAN be asked is MEMBERSHIP: every entry the oracle / the GPU path writes is one the reference
writes in some run.  Round 5's pin script re-ran the reference until it agreed (a retry loop); this script records the SET instead:

  * the judge's case: seed 424242, 40 000 read pairs of 4 genomes of 100-200 kb (plass_oracle synthreads = the GPU generator), the
    nucleotide chain of data/nuclassemble.sh for --iters iterations; the INPUT of every iteration is the oracle's (deterministic) chain;
  * per iteration the reference's `penguin kmermatcher` runs K times on that input (five at 8 threads, five at 1 by default); for every
    query the set of distinct entries seen is kept; a query with more than one is TIE-DEPENDENT, and so is each of its (query, target)
    LINES of which more than one version was seen;
  * the fixture holds, per iteration: the number of entries, a SHA-256 over all entries that are NOT tie-dependent (key order), and for
    every tie-dependent query the set of versions of every line — plus the entry the oracle wrote;
  * the script fails if a line the oracle writes is not among the reference's versions of that line (membership is asked line by line: an
    entry with m tied pairs has up to 2^m versions, of which K runs show a few).

tests/test_oracle_golden.py::test_oracle_strand_ties_are_a_reference_outcome and tests/test_gpu_parity.py::
test_nucleotide_strand_ties_are_a_reference_outcome regenerate the reads, follow the same chain and check the same two things.

    python tests/golden/make_strand_membership.py [--runs 8,8,8,8,8,1,1,1,1,1] [--iters 3]
"""
import argparse, hashlib, json, os, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from make_large_nucl import rest_db                      # noqa: E402

SYNTH = ["--pairs", "40000", "--seed", "424242", "--genomes", "4", "--genome-min-len", "100000", "--genome-max-len", "200000", "--abundance-sigma", "1.0"]


def entries(path):
    data = open(path, "rb").read()
    out = {}
    for line in open(path + ".index", "rb"):
        k, o, l = line.split()[:3]
        out[int(k)] = data[int(o):int(o) + int(l)]
    return out


def stable_sha(ent, skip):
    h = hashlib.sha256()
    for k in sorted(ent):
        if k not in skip:
            h.update(b"%d\x00" % k); h.update(ent[k])
    return h.hexdigest()


def by_target(entry):
    """prefilter entry -> {target key: line}"""
    out = {}
    for l in entry.decode("latin-1").split("\n"):
        if l and l != "\x00":
            out[int(l.split("\t", 1)[0])] = l
    return out


def line_sets(versions):
    """{entry bytes: runs} -> {target key: set of lines seen}"""
    out = {}
    for v in versions:
        for t, l in by_target(v).items():
            out.setdefault(t, set()).add(l)
    return out


def chain_step(orc, T, src, P, it, threads):
    """one iteration of data/nuclassemble.sh:99-137 by the oracle behind its kmermatcher: returns the next iteration's input"""
    run = lambda args: subprocess.run([orc] + args + ["--threads", str(threads)], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    run(["rescorediagonal", src, src, P("o_pref"), P("aln")] + T.NUCL_RS)
    run(["nuclassembleresults", src, P("aln"), P("assembly_%d" % it)] + T.NUCL_AS)
    run(["cyclecheck", P("assembly_%d" % it), P("cycle_%d" % it), "--max-seq-len", "200000", "--chop-cycle", "1"])
    rest_db(P("assembly_%d" % it), P("cycle_%d" % it), P("rest_%d" % it))
    return P("rest_%d" % it)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--runs", default="8,8,8,8,8,1,1,1,1,1", help="thread counts of the reference's kmermatcher runs per iteration")
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "strand_membership.json"))
    ap.add_argument("--report", default=os.path.join(ROOT, "profiles", "r06_strand_membership.txt"))
    a = ap.parse_args()
    import __graft_entry__ as g
    import conftest as T
    ref = os.path.join(os.environ.get("REF_BUILD", "/tmp/plass-build"), "src", "penguin")
    orc = g.oracle_bin()
    runs = [int(x) for x in a.runs.split(",")]
    rep = []

    def say(s):
        print(s, flush=True); rep.append(s)

    say("# strand-tied entries of the reference's nucleotide kmermatcher as SETS (tests/golden/make_strand_membership.py); reference: %s (unmodified)" % ref)
    say("# reads: plass_oracle synthreads %s; kmermatcher %s --max-seq-len 200000; %d reference runs per iteration (threads %s)" % (" ".join(SYNTH), " ".join(T.NUCL_KM), len(runs), a.runs))
    fx = {"made_by": "tests/golden/make_strand_membership.py (round 6): the UNMODIFIED reference's `penguin kmermatcher`, %d runs per iteration (threads %s), on the inputs of the oracle's chain" % (len(runs), a.runs),
          "synth": SYNTH, "kmermatcher": T.NUCL_KM + ["--max-seq-len", "200000"], "iterations": []}
    bad = 0
    with tempfile.TemporaryDirectory(dir=os.environ.get("TMPDIR", "/tmp")) as td:
        P = lambda n: os.path.join(td, n)
        subprocess.run([orc, "synthreads", P("reads")] + SYNTH, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        src = P("reads")
        for it in range(a.iters):
            km = T.NUCL_KM + ["--max-seq-len", "200000"]
            subprocess.run([orc, "kmermatcher", src, P("o_pref")] + km + ["--threads", "8"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            mine = entries(P("o_pref"))
            seen = {}                                          # key -> {entry bytes: runs}
            for r, th in enumerate(runs):
                subprocess.run([ref, "kmermatcher", src, P("r_pref")] + km + ["--threads", str(th), "-v", "1"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.STDOUT)
                e = entries(P("r_pref"))
                assert e.keys() == mine.keys(), "iteration %d run %d: the reference's key set differs from the oracle's" % (it, r)
                for k, v in e.items():
                    d = seen.setdefault(k, {})
                    d[v] = d.get(v, 0) + 1
                for f in (P("r_pref"), P("r_pref") + ".index", P("r_pref") + ".dbtype"):
                    os.remove(f)
            ties = sorted(k for k, d in seen.items() if len(d) > 1)
            # membership is asked LINE by line (one line = one (query, target) pair: "target\tscore\tdiagonal", the score's sign is the strand):
            # an entry with m tied pairs has up to 2^m versions, of which K runs show a few; every LINE the oracle writes must be one some run wrote
            rec = {"entries": len(mine), "tie_dependent_queries": len(ties), "tie_dependent_pairs": 0, "stable_sha256": stable_sha(mine, set(ties)), "ties": {}}
            outside = []
            for k in mine:
                if mine[k] in seen[k]:
                    continue
                lines = line_sets(seen[k])
                mo = by_target(mine[k])
                if mo.keys() != lines.keys() or any(mo[t] not in lines[t] for t in mo):
                    outside.append(k)
            bad += len(outside)
            for k in ties:
                lines = line_sets(seen[k])
                npairs = sum(1 for v in lines.values() if len(v) > 1)
                rec["tie_dependent_pairs"] += npairs
                rec["ties"][str(k)] = {"entry_versions_seen": len(seen[k]), "runs_per_version": sorted(seen[k].values(), reverse=True), "tied_pairs": npairs,
                                       "lines": {str(t): sorted(v) for t, v in sorted(lines.items())},
                                       "oracle": mine[k].decode("latin-1"), "oracle_entry_was_seen_whole": mine[k] in seen[k]}
            fx["iterations"].append(rec)
            say("iteration %d: %d entries, %d tie-dependent queries (the reference wrote more than one version of them in %d runs) with %d tie-dependent pairs: keys %s; "
                "oracle lines outside the reference's sets: in %d queries %s" % (it, len(mine), len(ties), len(runs), rec["tie_dependent_pairs"], ties[:12], len(outside), outside[:12]))
            for k in ties:
                vs = sorted(seen[k].items(), key=lambda x: -x[1])
                say("   query %d: %d versions of the entry over the runs (%s), %d of its %d pairs tie-dependent; the oracle's entry is version %s" %
                    (k, len(vs), ", ".join("%d run%s" % (n, "" if n == 1 else "s") for _, n in vs), rec["ties"][str(k)]["tied_pairs"], len(rec["ties"][str(k)]["lines"]),
                     [i for i, (v, _) in enumerate(vs) if v == mine[k]] or "none seen whole (every line of it was seen)"))
            src = chain_step(orc, T, src, P, it, 8)
    say("queries with an oracle line that NO reference run produced: %d" % bad)
    with open(a.out, "w") as f:
        json.dump(fx, f, indent=1); f.write("\n")
    with open(a.report, "w") as f:
        f.write("\n".join(rep) + "\n")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
