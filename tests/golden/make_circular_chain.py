#!/usr/bin/env python3
"""Makes tests/golden/circular_chain.json (VERDICT r5 item 5c / missing #4): a deep nucleotide chain whose community contains SMALL CIRCULAR
REPLICONS, so that `cyclecheck --chop-cycle 1` fires at depth — contigs that grew around a plasmid over several iterations until their ends
overlap — and not only on the crafted contigs of cyclecheck.tar.gz.

  reads()      numpy (PCG64, seed below): circular replicons of 2.5-9 kb at 30-60x and a few linear genomes of 20-40 kb at 15x; 150-nt reads of both
               strands, 0.2 % substitutions; reads of a circular replicon wrap around its origin.  Deterministic: the tests regenerate it.
  chain        data/nuclassemble.sh:99-137, ITERS iterations: kmermatcher -k 22 -> rescorediagonal -> nuclassembleresults -> cyclecheck --chop-cycle 1
               -> the non-circular rest; computed by the CPU oracle; the fixture holds entries / bytes / `dbsum` digest of every DB.
  --pin        (build container: REF_BUILD = /tmp/plass-build) every module call repeated by the UNMODIFIED reference on the SAME input DBs and the
               DBs compared entry for entry (the prefilter DB line by line as a member of the set of versions several reference runs write where
               the reference is not deterministic: make_strand_membership.py); the result is recorded in the fixture and in
               profiles/r06_circular_chain_pin.txt.

tests/test_oracle_golden.py::test_oracle_circular_chain (CPU) and tests/test_gpu_deep.py::test_circular_replicons_chop_cycle_at_depth (GPU)
follow the chain and compare every DB with the fixture.

    python tests/golden/make_circular_chain.py [--pin]
"""
import argparse, json, os, subprocess, sys, tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from make_large_nucl import db_sums, rest_db              # noqa: E402

SEED, ITERS, READ = 20260, 7, 150


def reads():
    """[n, 150] uint8 ASCII reads"""
    rng = np.random.default_rng(SEED)
    B = np.frombuffer(b"ACGT", dtype=np.uint8)
    comp = np.zeros(256, dtype=np.uint8); comp[B] = B[::-1]
    out = []
    genomes = [(int(rng.integers(2500, 9000)), True, float(rng.uniform(30, 60))) for _ in range(14)] + [(int(rng.integers(20000, 40000)), False, 15.0) for _ in range(3)]
    for L, circular, cov in genomes:
        g = B[rng.integers(0, 4, L)]
        n = int(L * cov / READ)
        src = np.concatenate([g, g[:READ]]) if circular else g
        starts = rng.integers(0, L if circular else L - READ + 1, n)
        r = src[starts[:, None] + np.arange(READ)[None, :]].copy()
        err = rng.random(r.shape) < 0.002
        r[err] = B[(np.searchsorted(B, r[err]) + rng.integers(1, 4, int(err.sum()))) % 4]
        rev = rng.random(n) < 0.5
        r[rev] = comp[r[rev][:, ::-1]]
        out.append(r)
    r = np.concatenate(out)
    return r[rng.permutation(len(r))]


def write_reads(path):
    r = reads()
    n = len(r)
    ent = np.empty((n, READ + 2), dtype=np.uint8); ent[:, :READ] = r; ent[:, READ] = 10; ent[:, READ + 1] = 0
    open(path, "wb").write(ent.tobytes())
    with open(path + ".index", "w") as f:
        for i in range(n):
            f.write("%d\t%d\t%d\n" % (i, i * (READ + 2), READ + 2))
    open(path + ".dbtype", "wb").write((1).to_bytes(4, "little"))
    return n


def entries(path):
    if os.path.exists(path):
        data = open(path, "rb").read()
    else:                                                     # the reference's per-thread data files, not merged yet
        data, i = b"", 0
        while os.path.exists("%s.%d" % (path, i)):
            data += open("%s.%d" % (path, i), "rb").read(); i += 1
    out = {}
    for line in open(path + ".index", "rb"):
        k, o, l = line.split()[:3]
        out[int(k)] = data[int(o):int(o) + int(l)]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pin", action="store_true", help="repeat every module call with the unmodified reference and compare (build container)")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "circular_chain.json"))
    a = ap.parse_args()
    import __graft_entry__ as g
    import conftest as T
    from make_strand_membership import by_target, line_sets
    orc = g.oracle_bin()
    ref = os.path.join(os.environ.get("REF_BUILD", "/tmp/plass-build"), "src", "penguin")
    rep = []

    def say(s):
        print(s, flush=True); rep.append(s)

    fx = {"made_by": "tests/golden/make_circular_chain.py: CPU oracle only (numpy read generator, seed %d)" % SEED, "seed": SEED, "iters": ITERS, "iterations": []}
    km = T.NUCL_KM + ["--max-seq-len", "200000"]
    run_o = lambda args: subprocess.run([orc] + args + ["--threads", "8"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    run_r = lambda args, th=8: subprocess.run([ref] + args + ["--threads", str(th), "-v", "1"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.STDOUT)
    n_ok = n_bad = 0
    with tempfile.TemporaryDirectory(dir=os.environ.get("TMPDIR", "/tmp")) as td:
        P = lambda n: os.path.join(td, n)
        n = write_reads(P("reads"))
        fx["reads"] = db_sums(P("reads"))
        say("# %d reads (seed %d): 14 circular replicons of 2.5-9 kb at 30-60x, 3 linear genomes of 20-40 kb at 15x" % (n, SEED))
        src = P("reads")
        for it in range(ITERS):
            p, al, asm, cyc, rest = P("pref"), P("aln"), P("assembly_%d" % it), P("cycle_%d" % it), P("rest_%d" % it)
            run_o(["kmermatcher", src, p] + km)
            run_o(["rescorediagonal", src, src, p, al] + T.NUCL_RS)
            run_o(["nuclassembleresults", src, al, asm] + T.NUCL_AS)
            run_o(["cyclecheck", asm, cyc, "--max-seq-len", "200000", "--chop-cycle", "1"])
            ncyc = rest_db(asm, cyc, rest)
            rec = {name: db_sums(path) for name, path in (("pref", p), ("aln", al), ("assembly", asm), ("cycle", cyc), ("rest", rest))}
            rec["n_cyclic"] = ncyc
            lens = [len(v) - 2 for v in entries(cyc).values()]
            rec["cycle_lengths"] = sorted(set(lens))
            say("iteration %d: %d sequences in, %d circular contigs taken out (%d distinct lengths, %s .. %s), longest contig %d" % (it, db_sums(src)["entries"], ncyc, len(set(lens)), min(lens) if lens else "-", max(lens) if lens else "-", max(len(v) - 2 for v in entries(asm).values())))
            if a.pin:
                # the reference on the SAME inputs, module by module
                seen = {}
                for th in (8, 8, 8, 1, 1):
                    run_r(["kmermatcher", src, P("r_pref")] + km, th)
                    for k, v in entries(P("r_pref")).items():
                        seen.setdefault(k, set()).add(v)
                mine = entries(p)
                out_of_set = [k for k in mine if mine[k] not in seen[k] and (by_target(mine[k]).keys() != line_sets(seen[k]).keys() or any(l not in line_sets(seen[k])[t] for t, l in by_target(mine[k]).items()))]
                ties = sum(1 for v in seen.values() if len(v) > 1)
                run_r(["rescorediagonal", src, src, p, P("r_aln")] + T.NUCL_RS)
                run_r(["nuclassembleresults", src, al, P("r_asm")] + T.NUCL_AS)
                run_r(["cyclecheck", asm, P("r_cyc"), "--max-seq-len", "200000", "--chop-cycle", "1"])
                same = {"pref (membership, 5 runs, %d tie-dependent entries)" % ties: not out_of_set, "aln": entries(al) == entries(P("r_aln")), "assembly": entries(asm) == entries(P("r_asm")), "cycle": entries(cyc) == entries(P("r_cyc"))}
                for name, ok in same.items():
                    n_ok += ok; n_bad += (not ok)
                    say("   %-8s iteration %d %s" % ("MATCH" if ok else "DIFFERS", it, name))
                rec["pinned_against_reference"] = all(same.values())
            fx["iterations"].append(rec)
            src = rest
    total = sum(r["n_cyclic"] for r in fx["iterations"])
    late = sum(r["n_cyclic"] for r in fx["iterations"][3:])
    say("circular contigs over the chain: %d, of them in iterations >= 3: %d" % (total, late))
    assert late > 0, "no circular contig at depth: the fixture would not test what it is for"
    if a.pin:
        fx["made_by"] += "; every DB of every iteration equals what the unmodified reference writes for the same input (--pin, profiles/r06_circular_chain_pin.txt: %d of %d module outputs)" % (n_ok, n_ok + n_bad)
        say("module outputs compared with the reference's: %d, identical (pref: member of the reference's set): %d, differing: %d" % (n_ok + n_bad, n_ok, n_bad))
        with open(os.path.join(ROOT, "profiles", "r06_circular_chain_pin.txt"), "w") as f:
            f.write("\n".join(rep) + "\n")
    with open(a.out, "w") as f:
        json.dump(fx, f, indent=1); f.write("\n")
    return 1 if n_bad else 0


if __name__ == "__main__":
    sys.exit(main())
