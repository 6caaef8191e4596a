"""Large-scale parity of PenguiN's nucleotide-level chains (BASELINE.json configs[4]; run with `-m gpu`): the nucleotide chain
(kmermatcher -k 22 -> rescorediagonal -> nuclassembleresults -> cyclecheck --chop-cycle) and the protein-guided chain (kmermatcher ->
rescorediagonal -a 1 -> proteinaln2nucl -> guidedassembleresults) on a multi-million-read sample of the community model with skewed
coverage, reads generated on the GPU, against checksums the CPU oracle produced for the same reads in the build container
(tests/golden/large_nucl.json, made by tests/golden/make_large_nucl.py).  This is where the libstdc++ heap replay, the memoised
posterior classes of the nucleotide comparator (and the pass loop that resolves classes on a threshold with the host libm), the
24-byte-record tiers and the long-contig tiers of cyclecheck meet queues and contigs the bundled example does not have."""
import json
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden", "large_nucl.json")


def dbsum(path):
    import __graft_entry__ as g
    out = subprocess.run([g.oracle_bin(), "dbsum", str(path)], stdout=subprocess.PIPE, check=True, text=True).stdout.strip().split("\t")
    f = dict(x.split("=") for x in out[1:])
    return {"entries": int(f["entries"]), "bytes": int(f["bytes"]), "digest": f["digest"]}


def check_file(path, want, what):
    got = dbsum(path)
    for k in ("entries", "bytes", "digest"):
        assert got[k] == want[k], "%s: %s differs from the CPU oracle's (got %s, expected %s)" % (what, k, got[k], want[k])
    for suffix in ("", ".index", ".dbtype"):
        if os.path.exists(str(path) + suffix):
            os.remove(str(path) + suffix)


def check_db(db, want, what):
    i = db.info()
    assert i["n"] == want["entries"], "%s: %d entries, the CPU oracle has %d" % (what, i["n"], want["entries"])
    assert db.digest() == (want["digest"], want["bytes"]), "%s: digest / bytes differ from the CPU oracle's (%s, expected %s)" % (what, db.digest(), (want["digest"], want["bytes"]))


@pytest.fixture(scope="module")
def gold():
    g = json.load(open(GOLD))
    assert 2 * g["pairs"] >= 2000000
    return g


@pytest.fixture(scope="module")
def ctx():
    import plass_amd
    c = plass_amd.Context(0)
    yield c
    c.close()


def _reads(ctx, gold):
    import bench
    sp = bench.synth_params(gold["config"], gold["pairs"])
    for k, v in gold["synth"].items():                       # the fixture was made for exactly these generator parameters
        assert getattr(sp, k) == pytest.approx(v), k
    reads, sst = ctx.synth_read_pairs(sp)
    assert sst.max_coverage > 3 * sst.mean_coverage
    check_db(reads, gold["reads"], "synthetic reads (GPU generator against the CPU generator)")
    return reads


@pytest.mark.timeout(1800)
def test_large_nucleotide_chain_against_oracle_checksums(ctx, gold, tmp_path):
    from test_gpu_parity import km_params, nucl_as_params
    import plass_amd
    db = _reads(ctx, gold)
    for it, want in enumerate(gold["nucl"]):
        cands, kst = ctx.kmermatcher(db, km_params(it, nucl=True))
        cands.write(tmp_path / "pref")
        check_file(tmp_path / "pref", want["pref"], "kmermatcher -k 22, iteration %d" % it)
        alns, _ = ctx.rescorediagonal(db, db, cands, plass_amd.RescoreParams(min_seq_id=0.99))
        cands.free()
        alns.write(tmp_path / "aln")
        check_file(tmp_path / "aln", want["aln"], "rescorediagonal (nucleotide), iteration %d" % it)
        out, ast = ctx.assembleresults(db, alns, nucl_as_params())
        alns.free(); db.free()
        check_db(out, want["assembly"], "nuclassembleresults, iteration %d" % it)
        cyc, rest, cst = ctx.cyclecheck(out, max_seq_len=200000, chop_cycle=True, with_rest=True)
        assert cst.n_cyclic == want["n_cyclic"]
        check_db(cyc, want["cycle"], "cyclecheck, iteration %d" % it)
        check_db(rest, want["rest"], "non-circular rest, iteration %d" % it)
        cyc.free(); out.free()
        db = rest
    db.free()


@pytest.mark.timeout(1800)
def test_large_guided_chain_against_oracle_checksums(ctx, gold, tmp_path):
    from test_gpu_parity import gd_km_params, gd_rs_params
    reads = _reads(ctx, gold)
    nu, aa = ctx.penguin_guided_inputs(reads)
    reads.free()
    check_db(nu, gold["guided_input"]["nucl"], "extractorfs x2 + concatdbs")
    check_db(aa, gold["guided_input"]["aa"], "translatenucs --add-orf-stop of the concatenated ORFs")
    for it, want in enumerate(gold["guided"]):
        cands, _ = ctx.kmermatcher(aa, gd_km_params())
        cands.write(tmp_path / "pref")
        check_file(tmp_path / "pref", want["pref"], "kmermatcher (guided), iteration %d" % it)
        alns, _ = ctx.rescorediagonal(aa, aa, cands, gd_rs_params())
        cands.free()
        alns.write(tmp_path / "aln")
        check_file(tmp_path / "aln", want["aln"], "rescorediagonal -a 1 (guided), iteration %d" % it)
        naln, _ = ctx.proteinaln2nucl(nu, aa, alns)
        alns.free()
        naln.write(tmp_path / "aln_nucl")
        check_file(tmp_path / "aln_nucl", want["aln_nucl"], "proteinaln2nucl, iteration %d" % it)
        nu2, aa2, _ = ctx.guidedassembleresults(nu, aa, naln)
        naln.free(); nu.free(); aa.free()
        check_db(nu2, want["nucl"], "guidedassembleresults (nucleotide ORFs), iteration %d" % it)
        check_db(aa2, want["aa"], "guidedassembleresults (protein twins), iteration %d" % it)
        nu, aa = nu2, aa2
    nu.free(); aa.free()
