import os
import subprocess
import sys
import tarfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_bin():
    """CPU oracle (test infrastructure) — built on demand with plain g++."""
    p = os.path.join(ROOT, "oracle", "build", "plass_oracle")
    if not os.path.exists(p) or not os.path.exists(os.path.join(ROOT, "oracle", "build", "liboracle.so")):
        subprocess.check_call(["make", "-j", "4"], cwd=os.path.join(ROOT, "oracle"))
    return p


@pytest.fixture(scope="session")
def golden(tmp_path_factory):
    """golden DBs written by the unmodified reference (tests/golden/make_golden.sh)"""
    d = tmp_path_factory.mktemp("golden")
    for name in ("example_aa.tar.gz", "example_aa_sweep.tar.gz", "example_nucl.tar.gz", "example_guided.tar.gz", "long_nucl.tar.gz", "adversarial.tar.gz", "stale_scan_cases.tar.gz", "findstart.tar.gz", "cyclecheck.tar.gz", "orfs.tar.gz", "concat_noncanonical.tar.gz", "strand_ties.tar.gz", "concat_preserve.tar.gz"):
        with tarfile.open(os.path.join(ROOT, "tests", "golden", name)) as t:
            t.extractall(d)
    return str(d)


def run_oracle(oracle_bin, args):
    p = subprocess.run([oracle_bin] + [str(a) for a in args], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert p.returncode == 0, p.stderr
    return p.stderr


def read_db(path):
    path = str(path)
    if os.path.exists(path):
        data = open(path, "rb").read()
    else:
        data, i = b"", 0
        while os.path.exists("%s.%d" % (path, i)):
            data += open("%s.%d" % (path, i), "rb").read()
            i += 1
    ent = {}
    for line in open(path + ".index", "rb"):
        k, o, l = line.split()[:3]
        assert int(k) not in ent
        ent[int(k)] = data[int(o):int(o) + int(l)]
    dbtype = int.from_bytes(open(path + ".dbtype", "rb").read()[:4], "little")
    return dbtype, ent


def assert_same_db(a, b, what=""):
    ta, ea = read_db(a)
    tb, eb = read_db(b)
    assert ta == tb, "%s: dbtype %d != %d" % (what, ta, tb)
    assert ea.keys() == eb.keys(), "%s: key sets differ" % what
    bad = [k for k in ea if ea[k] != eb[k]]
    if bad:
        _keep_failure(a, b, what)
    assert not bad, "%s: %d entries differ (keys %s), first key %d:\n%r\n%r" % (what, len(bad), bad[:10], bad[0], ea[bad[0]][:300], eb[bad[0]][:300])


def _keep_failure(a, b, what):
    """on a GPU box: keep both DBs of a parity failure under gpurun_out/ (merged back by gpurun) for offline analysis"""
    import glob
    import shutil
    out = os.path.join(ROOT, "gpurun_out", "parity_failures", "".join(c if c.isalnum() else "_" for c in what)[:60])
    try:
        os.makedirs(out, exist_ok=True)
        for tag, path in (("expected", str(a)), ("got", str(b))):
            for f in glob.glob(path + "*"):
                if os.path.getsize(f) < (8 << 20):
                    shutil.copy(f, os.path.join(out, tag + "_" + os.path.basename(f)))
    except OSError:
        pass


AA_KM = ["--alph-size", "13", "--kmer-per-seq", "60", "--kmer-per-seq-scale", "nucl:0.200,aa:0.000", "-k", "14", "-c", "0",
         "--cov-mode", "0", "--ignore-multi-kmer", "1"]
AA_RS = ["--rescore-mode", "3", "-e", "1e-05", "-c", "0", "-a", "0", "--cov-mode", "0", "--min-seq-id", "0.9"]
AA_AS = ["--min-seq-id", "0.9", "--max-seq-len", "65535", "--keep-target", "1", "--rescore-mode", "3"]
NUCL_KM = ["--alph-size", "5", "--kmer-per-seq", "60", "--kmer-per-seq-scale", "0.100", "-k", "22", "-c", "0", "--cov-mode", "0",
           "--ignore-multi-kmer", "1", "--hash-shift", "67", "--include-only-extendable", "1"]
NUCL_RS = ["--rescore-mode", "3", "-e", "1e-05", "-c", "0", "-a", "0", "--cov-mode", "0", "--min-seq-id", "0.99"]
NUCL_AS = ["--min-seq-id", "0.99", "--max-seq-len", "200000", "--keep-target", "1", "--rescore-mode", "3"]
# penguin guided_nuclassemble: protein k-mer matching / re-scoring on the translated ORFs, then nucleotide-level assembly
GD_KM = ["--alph-size", "nucl:5,aa:13", "--kmer-per-seq", "60", "--kmer-per-seq-scale", "0.100", "-k", "14", "-c", "0", "--cov-mode", "1",
         "--ignore-multi-kmer", "1", "--hash-shift", "67", "--include-only-extendable", "1"]
GD_RS = ["--rescore-mode", "3", "-e", "1e-05", "-c", "0", "-a", "1", "--cov-mode", "1", "--min-seq-id", "0.97"]
GD_P2N = ["--gap-open", "5", "--gap-extend", "2"]
GD_AS = ["--min-seq-id", "0.99", "--max-seq-len", "200000", "--keep-target", "1", "--rescore-mode", "3"]


def aa_iter_flags(i):
    return ["--hash-shift", "67" if i == 0 else "68", "--include-only-extendable", "0" if i == 0 else "1"]


def sweep_variants():
    """(name, module, flags) of tests/golden/example_aa_sweep.tar.gz (make_golden_sweep.sh): the reference run with non-default flags"""
    import io
    with tarfile.open(os.path.join(ROOT, "tests", "golden", "example_aa_sweep.tar.gz")) as t:
        txt = t.extractfile("sweep/variants.tsv").read().decode()
    out = []
    for line in io.StringIO(txt):
        name, mod, flags = line.rstrip("\n").split("\t")
        out.append((name, mod, flags.split()))
    return out


def sweep_positional(golden, mod, out):
    """positional arguments of a sweep variant; modules of the nucleotide / guided input sets are tagged "nucl:" / "guided:".
    Returns (module name, arguments, [(golden name suffix, output path)])"""
    out = str(out)
    if mod.startswith("nucl:"):
        s, m = os.path.join(golden, "nucl"), mod[5:]
        if m == "kmermatcher":
            return m, [f"{s}/seq_0", out], [("", out)]
        if m == "rescorediagonal":
            return m, [f"{s}/seq_0", f"{s}/seq_0", f"{s}/pref_0", out], [("", out)]
        return m, [f"{s}/seq_0", f"{s}/aln_0", out], [("", out)]
    if mod.startswith("guided:"):
        s, m = os.path.join(golden, "guided"), mod[7:]
        return m, [f"{s}/nucl_0", f"{s}/aa_0", f"{s}/aln_nucl_0", out, out + "_aa"], [("", out), ("_aa", out + "_aa")]
    s = os.path.join(golden, "aa")
    if mod == "kmermatcher":
        return mod, [f"{s}/seq_0", out], [("", out)]
    if mod == "rescorediagonal":
        return mod, [f"{s}/seq_0", f"{s}/seq_0", f"{s}/pref_0", out], [("", out)]
    return mod, [f"{s}/seq_0", f"{s}/aln_0", out], [("", out)]


def check_strand_membership(rec, ent, what):
    """tests/golden/strand_membership.json (make_strand_membership.py): the prefilter entries `ent` {key: bytes} of one nucleotide iteration
    against what the unmodified reference wrote in K runs — every query that is not tie-dependent byte for byte (one SHA-256 over them),
    every LINE of a tie-dependent query one of the versions some reference run wrote of it"""
    import hashlib
    ties = {int(k): v for k, v in rec["ties"].items()}
    assert len(ent) == rec["entries"], what
    h = hashlib.sha256()
    for k in sorted(ent):
        if k not in ties:
            h.update(b"%d\x00" % k); h.update(ent[k])
    assert h.hexdigest() == rec["stable_sha256"], "%s: an entry that does not depend on a strand tie differs from the reference's" % what
    for k, t in ties.items():
        mine = {}
        for l in ent[k].decode("latin-1").split("\n"):
            if l and l != "\x00":
                mine[l.split("\t", 1)[0]] = l
        assert mine.keys() == t["lines"].keys(), "%s: query %d names other targets than the reference" % (what, k)
        for tg, l in mine.items():
            assert l in t["lines"][tg], "%s: query %d, target %s: %r is not a line any of the reference's runs wrote (%r)" % (what, k, tg, l, t["lines"][tg])

