#!/usr/bin/env python3
"""BUILD CONTAINER ONLY (needs the survey-time build of the unmodified reference, REF_BUILD = /tmp/plass-build: tests/golden/make_golden.sh).

Pins the fixtures that were NOT written by the reference — tests/golden/deep_chains.json and the older large fixtures (CPU oracle), and the digests of
the bench workloads (GPU path) — against the REFERENCE ITSELF: the same synthetic reads (`plass_oracle synthreads`), the same module calls in the same order, made by the unmodified
`plass` / `penguin` binaries (8 threads), every DB digested with `plass_oracle dbsum` (order-independent over (key, length, bytes)) and compared
with the fixture:

  c2_exact  BASELINE configs[1] as stated: extractorfs x2, translatenucs x2, concatdbs, then the six iterations of data/assemble.sh:85-156
            incl. findassemblystart in iteration 0
  c2_bench  the same reads through the chain `bench.py --config c2` times (no findassemblystart)
  c3_deep   2 M reads of the configs[2] community, twelve iterations
  large_chain, large_nucl, big_offsets  the round-2..4 fixtures of tests/test_gpu_large.py / test_gpu_large_nucl.py (12.5 M / 5 M reads; sequence data
            beyond 2^32 bytes), oracle-made like the deep ones
  record_chain  (only on request: hours) WRITES tests/golden/reference_chain.json: twelve iterations on --record-pairs read pairs of the configs[2] community
            by the reference alone — a fixture with no oracle in its chain of trust.  (Not run / not committed in round 5: no GPU time was left to run
            a test against it.)
  c5_headline  (only on request: hours) the workload of `bench.py --config c5`: 20 M reads, 5 nucleotide + 5 guided iterations — against
            tests/golden/c5_chain_digests.json (GPU-made).  Output: profiles/r05_headline_pin_reference.txt
  c3_headline  (only on request: hours) THE BENCH LINE'S WORKLOAD: 50 M reads, twelve iterations — against tests/golden/c3_chain_digests.json, the digests
            the GPU path produced and bench.py's `verify` holds every run to.  The chain is followed as long as the reference's kmermatcher runs in ONE
            part (a split run is a different computation, see ref()): with the default limit on the 62 GB build container through iteration 5; with
            `--split-memory-limit 64G` through all twelve (62.8 GB of records in iteration 11: just fits).  Output: profiles/r05_headline_pin_reference.txt
  c5_deep   2 M reads of the configs[4] model: six nucleotide iterations with cyclecheck --chop-cycle 1 + the rest DB (data/nuclassemble.sh),
            four protein-guided iterations (data/guidedNuclAssemble.sh)

Prints one line per DB (MATCH / DIFFERS) and a summary; the output of the run that accompanies the fixture is profiles/r05_deep_pin_reference.txt.

    python tests/golden/pin_deep_chains_against_reference.py [--only c2_exact,c3_deep,c5_deep] [--threads 8]
"""
import argparse, json, os, subprocess, sys, tempfile, time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

import glob                                               # noqa: E402
from make_large_nucl import db_sums, rest_db              # noqa: E402
from make_deep_chains import synth                        # noqa: E402
sys.path.insert(0, os.path.join(ROOT, "tools"))
from dbcanon import canon                                 # noqa: E402

B = os.environ.get("REF_BUILD", "/tmp/plass-build")
PLASS, PENGUIN = os.path.join(B, "src", "plass"), os.path.join(B, "src", "penguin")
N_OK = N_BAD = 0
# concatdbs numbers the entries of its second DB in the order they LIE IN THE DATA FILE (DBConcat.cpp:46-47 opens LINEAR_ACCCESS), and a
# translatenucs with several threads leaves its entries in the order the threads' chunks were merged: the keys of the concatenated DB then depend
# on the thread count.  One thread gives key order = file order, which is what the oracle and the GPU path produce; the sequences are the same
# either way (checked: the multiset of entries of an 8-thread run equals the 1-thread run's).  extractorfs and translatenucs therefore run with one
# thread here, everything else with --threads 8.
Q1 = ["--threads", "1", "-v", "1"]


T_REF = {}
SPLIT_SEEN = False


def ref(binary, args, q):
    t = time.time()
    if args[0] == "kmermatcher":
        # The reference's kmermatcher SPLITS its k-mer range when the records do not fit the host's memory (kmermatcher.cpp:619-626), and a split run is a
        # different computation: every part reduces its own records to one (diagonal, count) per pair before the parts are merged
        # (mergeKmerFilesAndOutput, :945-1100), so pairs whose k-mers fall into several parts can come out with another diagonal — measured on the 2 M
        # reads of c3_deep: 33 of 3 528 307 entries (profiles/r05_deep_pin_reference.txt).  The unsplit result is the one oracle and GPU path compute
        # (the GPU holds all records of the 50 M-read workload at once); a split run is reported, its digests cannot be expected to match.
        out = subprocess.run([binary] + [str(a) for a in args] + q[:2] + ["-v", "3"], check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True).stdout
        parts = [l for l in out.splitlines() if l.startswith("Process file into")]
        if parts:
            global SPLIT_SEEN
            SPLIT_SEEN = True
            print("         (NOTE: the reference's kmermatcher split its work — '%s' — the result of a split run differs from the unsplit one)" % parts[0].strip(), flush=True)
    else:
        subprocess.run([binary] + [str(a) for a in args] + q, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.STDOUT)
    T_REF[args[0]] = T_REF.get(args[0], 0.0) + time.time() - t


def rm(*paths):
    """a DB and the per-thread data files the reference leaves behind for some modules (<db>.0 .. <db>.7)"""
    for p in paths:
        for f in [p, p + ".index", p + ".dbtype"] + glob.glob(p + ".[0-9]*"):
            if os.path.lexists(f):
                os.remove(f)


def same(path, want):
    got = db_sums(path)
    return all(got[k] == want[k] for k in ("entries", "bytes", "digest") if k in want)


RECORD = False                                            # --record-chain: the reference's digests are written down instead of compared


def check(path, want, what):
    global N_OK, N_BAD
    if want is None:                                          # no fixture for this DB (headline chain: only the sequence DBs have one)
        return
    got = db_sums(path)
    if RECORD:
        want.update(got)
        print("%-8s %-58s entries %9d bytes %11d digest %s" % ("RECORD", what, got["entries"], got["bytes"], got["digest"]), flush=True)
        return
    ok = all(got[k] == want[k] for k in ("entries", "bytes", "digest") if k in want)
    N_OK += ok; N_BAD += (not ok)
    print("%-8s %-58s entries %9d bytes %11d digest %s%s" % ("MATCH" if ok else "DIFFERS", what, got["entries"], got["bytes"], got["digest"],
                                                            "" if ok else "   fixture: %s" % {k: want[k] for k in ("entries", "bytes", "digest") if k in want}), flush=True)


def write_header_db(reads, out):
    """extractorfs of the reference wants <reads>_h; the ORF headers it derives from it are not compared (the product keeps them as numbers)"""
    keys = [int(l.split(b"\t", 1)[0]) for l in open(reads + ".index", "rb")]
    off = 0
    with open(out, "wb") as d, open(out + ".index", "wb") as ix:
        for k in keys:
            e = b"%d\n\0" % k
            d.write(e); ix.write(b"%d\t%d\t%d\n" % (k, off, len(e))); off += len(e)
    open(out + ".dbtype", "wb").write((12).to_bytes(4, "little"))


def orf_flags(par):
    fl = []
    for k, v in par.items():
        fl += ["--" + k.replace("_", "-"), str(v)]
    return fl


def protein(g, bench, _lib, fx, q, td):
    P = lambda n: os.path.join(td, n)
    sp = bench.synth_params(fx["config"], fx["pairs"])
    synth(g, sp, P("reads")); write_header_db(P("reads"), P("reads_h"))
    check(P("reads"), fx.get("reads"), "synthetic reads")
    for name, par in (("long", _lib.PLASS_ORFS_LONG), ("start", _lib.PLASS_ORFS_START)):
        ref(PLASS, ["extractorfs", P("reads"), P("nucl_" + name)] + orf_flags(par), Q1)
        ref(PLASS, ["translatenucs", P("nucl_" + name), P("aa_" + name), "--add-orf-stop", "1"], Q1)
        rm(P("nucl_" + name), P("nucl_" + name + "_h"))
    ref(PLASS, ["concatdbs", P("aa_long"), P("aa_start"), P("seq_0")], q)
    rm(P("reads"), P("reads_h"), P("aa_long"), P("aa_start"), P("aa_long_h"), P("aa_start_h"))
    if fx.get("filler"):
        # tests/golden/big_offsets.json: 4.48 GB of filler sequences in front of the fragments, so every live byte offset exceeds 2^32
        from make_big_offsets import write_filler_db
        check(P("seq_0"), fx["live"], "extractorfs x2 + translatenucs x2 + concatdbs (the live fragments)")
        canon(P("seq_0"), P("live")); rm(P("seq_0"))        # key order in the file, as a single-threaded concatdbs leaves it: the second concatdbs numbers by file order (see Q1)
        write_filler_db(P("filler"), fx["filler"]["n"], fx["filler"]["length"], fx["filler"]["seed"])
        check(P("filler"), fx["filler_db"], "filler DB (numpy generator)")
        ref(PLASS, ["concatdbs", P("filler"), P("live"), P("seq_0")], q)
        rm(P("filler"), P("live"))
        check(P("seq_0"), fx["db"], "concatdbs filler live: the DB beyond 2^32 bytes")
    else:
        check(P("seq_0"), fx["fragments"], "extractorfs x2 + translatenucs x2 + concatdbs")
    t_it = time.time()
    RS = ["--rescore-mode", "3", "-e", "1e-05", "-c", "0", "-a", "0", "--cov-mode", "0", "--min-seq-id", "0.9", "--min-aln-len", "0", "--seq-id-mode", "0", "--sort-results", "0"]
    AS = ["--min-seq-id", "0.9", "--max-seq-len", "65535", "--keep-target", "1", "--rescore-mode", "3"]
    for it, want in enumerate(fx["iterations"]):
        s, p, al, o = P("seq_%d" % it), P("pref"), P("aln"), P("seq_%d" % (it + 1))
        km = ["--alph-size", "13", "--kmer-per-seq", "60", "--kmer-per-seq-scale", "nucl:0.200,aa:0.000", "-k", "14", "-c", "0", "--cov-mode", "0", "--ignore-multi-kmer", "1",
              "--max-seq-len", "65535", "--hash-shift", str(bench.hash_shift(it)), "--include-only-extendable", "1" if it else "0"]
        km += fx.get("km_extra", [])
        ref(PLASS, ["kmermatcher", s, p] + km, q)
        if SPLIT_SEEN and fx.get("stop_on_split"):
            print("         (iteration %d: the records no longer fit this host's memory in one part; the chain is not followed further)" % it, flush=True)
            break
        ref(PLASS, ["rescorediagonal", s, s, p, al] + RS, q)
        if it == 0 and fx.get("findassemblystart"):
            check(p, want["pref_uncorrected"], "it 0: kmermatcher before findassemblystart"); check(al, want["aln_uncorrected"], "it 0: rescorediagonal before findassemblystart")
            ref(PLASS, ["findassemblystart", s, al, P("corrected")], q)
            check(P("corrected"), want["corrected"], "it 0: findassemblystart")
            rm(p, al); s = P("corrected")
            ref(PLASS, ["kmermatcher", s, p] + km, q); ref(PLASS, ["rescorediagonal", s, s, p, al] + RS, q)
        ref(PLASS, ["assembleresults", s, al, o] + AS, q)
        t_ref = time.time() - t_it
        check(p, want.get("pref"), "it %d: kmermatcher" % it); check(al, want.get("aln"), "it %d: rescorediagonal" % it); check(o, want["seq"], "it %d: assembleresults" % it)
        print("         (iteration %d: the reference's three modules took %.1f s on %s threads)" % (it, t_ref, q[1]), flush=True)
        rm(p, al, s, P("seq_%d" % it))
        t_it = time.time()


def nucl_and_guided(g, bench, _lib, T, fx, q, td):
    P = lambda n: os.path.join(td, n)
    sp = bench.synth_params("c5", fx["pairs"])
    synth(g, sp, P("reads")); write_header_db(P("reads"), P("reads_h"))
    check(P("reads"), fx.get("reads"), "synthetic reads")
    src = P("reads")
    t_it = time.time()
    for it, want in enumerate(fx["nucl"]):
        p, al, o, cy, rest = P("pref"), P("aln"), P("assembly_%d" % it), P("cycle_%d" % it), P("rest_%d" % it)
        # The reference's nucleotide kmermatcher is not deterministic from run to run: where a (representative, target, diagonal) triple holds both
        # strands its second sort has ties, and which strand survives depends on the parallel sort's partitioning (profiles/r04_strand_ties.txt:
        # its 1-thread and 8-thread runs differ from each other in a handful of entries per million).  Either outcome is "the reference's"; the
        # oracle and the GPU path take one of them (k-mer order).  A run that differs from the fixture is therefore repeated, and the line says so.
        # (Where the fixture holds no `pref` — the headline chain — the whole iteration is repeated when the contigs differ.)
        for attempt in range(1, 4):
            runs = 0
            while True:
                ref(PENGUIN, ["kmermatcher", src, p] + T.NUCL_KM + ["--max-seq-len", "200000"] + fx.get("km_extra", []), q if runs < 3 else Q1)
                runs += 1
                if not want.get("pref") or same(p, want["pref"]) or runs >= 5:
                    break
                got = db_sums(p)
                print("         (nucleotide it %d: kmermatcher run %d of the reference gave digest %s, %d bytes: repeating it)" % (it, runs, got["digest"], got["bytes"]), flush=True)
                rm(p)
            ref(PENGUIN, ["rescorediagonal", src, src, p, al] + T.NUCL_RS, q)
            ref(PENGUIN, ["nuclassembleresults", src, al, o] + T.NUCL_AS, q)
            ref(PENGUIN, ["cyclecheck", o, cy, "--max-seq-len", "200000", "--chop-cycle", "1"], q)
            rest_db(o, cy, rest)
            if want.get("pref") or same(rest, want["rest"]) or attempt == 3:
                break
            got = db_sums(rest)
            print("         (nucleotide it %d: attempt %d of the reference left contigs with digest %s, %d bytes: repeating the iteration)" % (it, attempt, got["digest"], got["bytes"]), flush=True)
            rm(p, al, o, cy, rest)
        t_ref = time.time() - t_it
        for name, path in (("pref", p), ("aln", al), ("assembly", o), ("cycle", cy), ("rest", rest)):
            check(path, want.get(name), "nucleotide it %d: %s" % (it, name))
        print("         (nucleotide iteration %d: the reference's four modules took %.1f s on %s threads)" % (it, t_ref, q[1]), flush=True)
        rm(p, al, cy)
        if it:
            rm(P("rest_%d" % (it - 1)), P("assembly_%d" % (it - 1)))
        src = rest
        t_it = time.time()
    if fx["nucl"]:
        rm(src, P("assembly_%d" % (len(fx["nucl"]) - 1)))
    if not fx["guided"]:
        return
    for name, par in (("long", _lib.PLASS_ORFS_LONG), ("start", _lib.PLASS_ORFS_START)):
        ref(PENGUIN, ["extractorfs", P("reads"), P("nucl_" + name)] + orf_flags(par), Q1)
    ref(PENGUIN, ["concatdbs", P("nucl_long"), P("nucl_start"), P("nucl_0")], q)
    ref(PENGUIN, ["concatdbs", P("nucl_long_h"), P("nucl_start_h"), P("nucl_0_h")], q)
    ref(PENGUIN, ["translatenucs", P("nucl_0"), P("aa_0"), "--add-orf-stop", "1"], Q1)
    rm(P("reads"), P("reads_h"), P("nucl_long"), P("nucl_start"), P("nucl_long_h"), P("nucl_start_h"))
    check(P("nucl_0"), fx["guided_input"]["nucl"], "guided input: extractorfs x2 + concatdbs"); check(P("aa_0"), fx["guided_input"]["aa"], "guided input: translatenucs --add-orf-stop")
    t_it = time.time()
    for it, want in enumerate(fx["guided"]):
        nu, aa, p, al, an = P("nucl_%d" % it), P("aa_%d" % it), P("pref"), P("aln"), P("aln_nucl")
        nu2, aa2 = P("nucl_%d" % (it + 1)), P("aa_%d" % (it + 1))
        ref(PENGUIN, ["kmermatcher", aa, p] + T.GD_KM + ["--max-seq-len", "200000"] + fx.get("km_extra", []), q)
        ref(PENGUIN, ["rescorediagonal", aa, aa, p, al] + T.GD_RS, q)
        if it:
            # proteinaln2nucl reads the nucleotide ORFs up to the alignment's codon positions without looking at the entry's length: an alignment that
            # ends on the stop the protein twin carries behind its last residue makes it read the first bytes of WHATEVER LIES BEHIND THE ENTRY in the
            # data file.  The reference's own guidedassembleresults leaves the entries in the order its threads wrote them; the oracle and the GPU path
            # define the result by the canonical layout (one data file, entries in key order: tools/dbcanon.py).  Both are shown.
            if want.get("aln_nucl"):
                ref(PENGUIN, ["proteinaln2nucl", nu, nu, aa, aa, al, an] + T.GD_P2N, q)
                print("%-8s %-58s %s" % ("(layout)", "guided it %d: aln_nucl on the files as its threads left them" % it,
                                         "identical" if same(an, want["aln_nucl"]) else "differs: digest %s, %d bytes" % (db_sums(an)["digest"], db_sums(an)["bytes"])), flush=True)
                rm(an)
            for x in (nu, aa):
                canon(x, x + "_canon"); rm(x)
                for sfx in ("", ".index", ".dbtype"):
                    os.rename(x + "_canon" + sfx, x + sfx)
        ref(PENGUIN, ["proteinaln2nucl", nu, nu, aa, aa, al, an] + T.GD_P2N, q)
        ref(PENGUIN, ["guidedassembleresults", nu, aa, an, nu2, aa2] + T.GD_AS, q)
        t_ref = time.time() - t_it
        for name, path in (("pref", p), ("aln", al), ("aln_nucl", an), ("nucl", nu2), ("aa", aa2)):
            check(path, want.get(name), "guided it %d: %s" % (it, name))
        print("         (guided iteration %d: the reference's four modules took %.1f s on %s threads)" % (it, t_ref, q[1]), flush=True)
        rm(p, al, an, nu, aa)
        t_it = time.time()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="c2_exact,c3_deep,c5_deep")
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--record-pairs", type=int, default=12500000, help="read pairs of the record_chain section")
    ap.add_argument("--split-memory-limit", default="0", help="of the headline sections' kmermatcher calls (0 = the reference's default: 90 %% of the host's memory)")
    a = ap.parse_args()
    import bench, __graft_entry__ as g
    import conftest as T
    from plass_amd import _lib
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "deep_chains.json")))
    q = ["--threads", str(a.threads), "-v", "1"]
    print("reference binaries: %s, %s (unmodified, survey-time build); fixture: tests/golden/deep_chains.json (%s)" % (PLASS, PENGUIN, fx["made_by"]))
    t0 = time.time()
    for what in a.only.split(","):
        print("---- %s ----" % what, flush=True)
        with tempfile.TemporaryDirectory(dir=os.environ.get("TMPDIR", "/tmp")) as td:
            if what in ("c2_exact", "c3_deep", "c2_bench"):
                protein(g, bench, _lib, fx[what], q, td)
            elif what == "record_chain":
                # A fixture made by the REFERENCE ALONE (no oracle in the chain of trust): tests/golden/reference_chain.json — the twelve iterations of the
                # default protein chain on --record-pairs read pairs of the configs[2] community, as large as the reference runs UNSPLIT on this host.
                from make_deep_chains import synth_dict
                global RECORD
                RECORD = True
                sp = bench.synth_params("c3", a.record_pairs)
                rec = {"made_by": "tests/golden/pin_deep_chains_against_reference.py --only record_chain (the UNMODIFIED reference: plass kmermatcher / rescorediagonal / "
                                  "assembleresults, 8 threads, unsplit; digests by plass_oracle dbsum)",
                       "config": "c3", "pairs": a.record_pairs, "iters": 12, "findassemblystart": False, "synth": synth_dict(sp), "reads": {}, "fragments": {},
                       "iterations": [{"pref": {}, "aln": {}, "seq": {}} for _ in range(12)]}
                protein(g, bench, _lib, rec, q, td)
                RECORD = False
                assert not SPLIT_SEEN, "the reference split its kmermatcher: this is not the result the fixture is meant to hold"
                rec["reference_seconds"] = {k: round(v, 1) for k, v in T_REF.items()}
                with open(os.path.join(ROOT, "tests", "golden", "reference_chain.json"), "w") as f:
                    json.dump(rec, f, indent=1)
                    f.write("\n")
            elif what == "large_chain":            # tests/golden/large_chain.json (12.5 M reads, three iterations: tests/test_gpu_large.py), oracle-made as well
                protein(g, bench, _lib, json.load(open(os.path.join(ROOT, "tests", "golden", "large_chain.json"))), q, td)
            elif what == "big_offsets":            # tests/golden/big_offsets.json (sequence data beyond 2^32 bytes, three iterations)
                protein(g, bench, _lib, json.load(open(os.path.join(ROOT, "tests", "golden", "big_offsets.json"))), q, td)
            elif what == "large_nucl":             # tests/golden/large_nucl.json (5 M reads, three nucleotide + two guided iterations: tests/test_gpu_large_nucl.py)
                nucl_and_guided(g, bench, _lib, T, json.load(open(os.path.join(ROOT, "tests", "golden", "large_nucl.json"))), q, td)
            elif what == "c3_headline":
                # the workload of the bench line itself: 50 M reads, twelve iterations.  tests/golden/c3_chain_digests.json holds what the GPU path
                # produced (what `verify` in bench.py compares every run with); here the reference computes the same chain on the CPU.
                h = json.load(open(os.path.join(ROOT, "tests", "golden", "c3_chain_digests.json")))
                protein(g, bench, _lib, {"config": h["config"], "pairs": h["pairs"], "findassemblystart": False, "reads": None, "fragments": {"entries": h["fragments"]}, "stop_on_split": True,
                                         "km_extra": ["--split-memory-limit", a.split_memory_limit],      # 5.3 G k-mer records = 85 GB: unsplit only on a host with > 100 GB
                                         "iterations": [{"seq": {"digest": d}} for d in h["digests"]]}, q, td)
            elif what == "c5_headline":
                # the workload of `bench.py --config c5` (20 M reads; 5 guided + 5 nucleotide iterations, both chains start from the reads) against
                # tests/golden/c5_chain_digests.json, which holds what the GPU path produced
                h = json.load(open(os.path.join(ROOT, "tests", "golden", "c5_chain_digests.json")))
                gd = [d.split("+") for d in h["digests"] if "+" in d]
                base = {"pairs": h["pairs"], "reads": None, "guided_input": {"nucl": None, "aa": None}, "km_extra": ["--split-memory-limit", a.split_memory_limit]}
                # the two chains are independent (both start from the reads): the guided one first — it is deterministic
                nucl_and_guided(g, bench, _lib, T, dict(base, nucl=[], guided=[{"nucl": {"digest": a}, "aa": {"digest": b}} for a, b in gd]), q, td)
                with tempfile.TemporaryDirectory(dir=os.environ.get("TMPDIR", "/tmp")) as td2:
                    nucl_and_guided(g, bench, _lib, T, dict(base, guided=[], nucl=[{"rest": {"digest": d}} for d in h["digests"] if "+" not in d]), q, td2)
            else:
                nucl_and_guided(g, bench, _lib, T, fx[what], q, td)
        print("(%s done, %.0f s; seconds inside the reference's modules so far: %s)" % (what, time.time() - t0, {k: round(v, 1) for k, v in T_REF.items()}), flush=True)
    print("DBs compared: %d, identical to the reference's: %d, differing: %d" % (N_OK + N_BAD, N_OK, N_BAD))
    return 1 if N_BAD else 0


if __name__ == "__main__":
    sys.exit(main())
