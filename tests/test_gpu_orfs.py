"""GPU parity tests of row N2 (`-m gpu`): extractorfs / translatenucs / concatdbs through the C-ABI against DBs written by the
unmodified reference (tests/golden/orfs.tar.gz; the bundled example's preprocessing chain that must reproduce aa/seq_0),
and against the CPU oracle on reads made by the GPU read generator (include/plasship_synth.h).  Bit-exact."""
import os

import numpy as np
import pytest

from conftest import assert_same_db, read_db, run_oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import plass_amd
    c = plass_amd.Context(0)
    yield c
    c.close()


def orf_params(flags):
    """reference flag list -> OrfParams"""
    import plass_amd
    kw = {}
    names = {"--min-length": "min_length", "--max-length": "max_length", "--max-gaps": "max_gaps", "--contig-start-mode": "contig_start_mode",
             "--contig-end-mode": "contig_end_mode", "--orf-start-mode": "orf_start_mode", "--translation-table": "translation_table"}
    for k, v in zip(flags[0::2], flags[1::2]):
        if k in names:
            kw[names[k]] = int(v)
        elif k == "--forward-frames":
            kw["forward_frames"] = v
        elif k == "--reverse-frames":
            kw["reverse_frames"] = v
        elif k == "--translate":
            kw["translate"] = bool(int(v))
        elif k == "--use-all-table-starts":
            kw["use_all_table_starts"] = bool(int(v))
        else:
            raise AssertionError("unknown flag " + k)
    return plass_amd.OrfParams(**kw)


def test_golden_extractorfs_translatenucs(ctx, golden, tmp_path):
    """hostile reads (IUPAC, N, lower case, U/u, illegal characters, tiny sequences), the reference's four flag sets: ORF DB,
    header DB, translations with and without --add-orf-stop"""
    o = os.path.join(golden, "orf")
    reads = ctx.read_seqdb(f"{o}/in")
    flags = [l.split() for l in open(os.path.join(o, "FLAGS")).read().splitlines() if l.strip()]
    for n, fl in zip((1, 2, 4, 5), flags):
        orfs, hdr, st = ctx.extractorfs(reads, orf_params(fl))
        orfs.write(tmp_path / f"orfs_{n}"); hdr.write(str(tmp_path / f"orfs_{n}") + "_h")
        assert_same_db(f"{o}/orfs_{n}", tmp_path / f"orfs_{n}", f"extractorfs flag set {n}")
        assert_same_db(f"{o}/orfs_{n}_h", str(tmp_path / f"orfs_{n}") + "_h", f"extractorfs headers, flag set {n}")
        assert st.n_out == orfs.info()["n"] == hdr.count()
        aa, _ = ctx.translatenucs(orfs, hdr, add_orf_stop=True)
        aa.write(tmp_path / f"aa_stop_{n}")
        assert_same_db(f"{o}/aa_stop_{n}", tmp_path / f"aa_stop_{n}", f"translatenucs --add-orf-stop 1, flag set {n}")
        aa2, _ = ctx.translatenucs(orfs, None, add_orf_stop=False)
        aa2.write(tmp_path / f"aa_{n}")
        assert_same_db(f"{o}/aa_{n}", tmp_path / f"aa_{n}", f"translatenucs --add-orf-stop 0, flag set {n}")
        # the same through the DB files (what the command line does): header DB parsed from text
        orfs_f = ctx.read_seqdb(tmp_path / f"orfs_{n}"); hdr_f = ctx.read_orfhdr(str(tmp_path / f"orfs_{n}") + "_h")
        aa3, _ = ctx.translatenucs(orfs_f, hdr_f, add_orf_stop=True)
        aa3.write(tmp_path / f"aa_stop_file_{n}")
        assert_same_db(f"{o}/aa_stop_{n}", tmp_path / f"aa_stop_file_{n}", f"translatenucs from DB files, flag set {n}")
        for x in (aa, aa2, aa3, orfs_f, hdr_f, orfs, hdr):
            x.free()
    small_reads = ctx.read_seqdb(f"{o}/in_small")
    small = [l.split() for l in open(os.path.join(o, "FLAGS_SMALL")).read().splitlines() if l.strip()]     # any-to-stop; --translate 1
    for name, fl in zip(("small_orfs_any", "small_orfs_translated"), small):
        orfs, hdr, _ = ctx.extractorfs(small_reads, orf_params(fl))
        orfs.write(tmp_path / name); hdr.write(str(tmp_path / name) + "_h")
        assert_same_db(f"{o}/{name}", tmp_path / name, f"extractorfs {name}")
        assert_same_db(f"{o}/{name}_h", str(tmp_path / name) + "_h", f"extractorfs {name} headers")


def test_golden_preprocessing_chain_reproduces_iteration0_input(ctx, golden, tmp_path):
    """nucl_reads of the bundled example -> two extractorfs passes -> translatenucs --add-orf-stop -> concatdbs must be the protein
    fragment DB the hot path's golden vectors start from (aa/seq_0), data/assemble.sh:41-77"""
    reads = ctx.read_seqdb(os.path.join(golden, "nucl", "seq_0"))
    frag = ctx.plass_fragments(reads)
    frag.write(tmp_path / "aa_6f_start_long")
    assert_same_db(os.path.join(golden, "aa", "seq_0"), tmp_path / "aa_6f_start_long", "preprocessing chain -> aa_6f_start_long")


def test_illegal_parameter_combination_fails_loudly(ctx, golden):
    import plass_amd
    reads = ctx.read_seqdb(os.path.join(golden, "orf", "in_small"))
    with pytest.raises(plass_amd.PlasshipError):
        ctx.extractorfs(reads, plass_amd.OrfParams(orf_start_mode=1, contig_start_mode=1))
    with pytest.raises(plass_amd.PlasshipError):
        ctx.extractorfs(reads, plass_amd.OrfParams(translation_table=4))


def test_synth_reads_deterministic_and_chain_vs_oracle(ctx, oracle_bin, tmp_path):
    """GPU read generator: same seed -> same bytes, other seed -> other bytes, pairs are consistent; the preprocessing chain on
    20 000 generated reads equals the oracle's, module by module"""
    import plass_amd
    par = plass_amd.SynthParams(n_pairs=10000, seed=5, n_genomes=6, genome_min_len=40000, genome_max_len=90000, abundance_sigma=1.0)
    r1, st = ctx.synth_read_pairs(par)
    r2, _ = ctx.synth_read_pairs(par)
    d1, o1, e1, k1 = r1.download(); d2 = r2.download()[0]
    assert d1 == d2 and len(k1) == 20000 and np.all(e1 == 152) and np.array_equal(k1, np.arange(20000, dtype=np.uint32))
    assert set(d1) <= set(b"ACGT\n\x00") and st.n_genes > 50 and st.genome_bases >= 6 * 40000
    r3, _ = ctx.synth_read_pairs(plass_amd.SynthParams(n_pairs=10000, seed=6, n_genomes=6, genome_min_len=40000, genome_max_len=90000, abundance_sigma=1.0))
    assert r3.download()[0] != d1
    r1.write(tmp_path / "reads")
    common = ["--max-gaps", "0", "--orf-start-mode", "0", "--forward-frames", "1,2,3", "--reverse-frames", "1,2,3", "--translation-table", "1",
              "--translate", "0", "--use-all-table-starts", "0"]
    run_oracle(oracle_bin, ["extractorfs", tmp_path / "reads", tmp_path / "o_start", "--min-length", "20", "--max-length", "45", "--contig-start-mode", "1",
                            "--contig-end-mode", "0"] + common)
    run_oracle(oracle_bin, ["extractorfs", tmp_path / "reads", tmp_path / "o_long", "--min-length", "45", "--max-length", "32734", "--contig-start-mode", "2",
                            "--contig-end-mode", "2"] + common)
    for n in ("start", "long"):
        run_oracle(oracle_bin, ["translatenucs", tmp_path / f"o_{n}", tmp_path / f"o_aa_{n}", "--translation-table", "1", "--add-orf-stop", "1"])
    run_oracle(oracle_bin, ["concatdbs", tmp_path / "o_aa_long", tmp_path / "o_aa_start", tmp_path / "o_frag"])
    for name, kw in (("long", plass_amd._lib.PLASS_ORFS_LONG), ("start", plass_amd._lib.PLASS_ORFS_START)):
        orfs, hdr, _ = ctx.extractorfs(r1, plass_amd.OrfParams(**kw))
        orfs.write(tmp_path / f"g_{name}"); hdr.write(str(tmp_path / f"g_{name}") + "_h")
        assert_same_db(tmp_path / f"o_{name}", tmp_path / f"g_{name}", f"extractorfs {name} vs oracle")
        assert_same_db(str(tmp_path / f"o_{name}") + "_h", str(tmp_path / f"g_{name}") + "_h", f"extractorfs {name} headers vs oracle")
    frag = ctx.plass_fragments(r1)
    frag.write(tmp_path / "g_frag")
    assert_same_db(tmp_path / "o_frag", tmp_path / "g_frag", "preprocessing chain vs oracle")
    i = frag.info()
    assert i["n"] > 20000 and i["dbtype"] == 0
    # the fragments are usable by the hot path right away (device-built DB, no host index yet)
    cands, kst = ctx.kmermatcher(frag, plass_amd.KmermatchParams(hash_shift=67, include_only_extendable=False))
    assert kst.n_candidates > 0


def test_concatdbs_numbers_b_in_file_order(ctx, golden, tmp_path):
    """concatdbs on DBs whose data files are in thread order (written by the reference's translatenucs on 8 threads, raw files): the
    handles keep every entry's rank in the data file, and B's new keys follow it like the reference's LINEAR_ACCCESS reader
    (DBConcat.cpp:46-47,113-118); through the C-ABI and through the command line"""
    import subprocess
    c = os.path.join(golden, "concat")
    a = ctx.read_seqdb(f"{c}/aaA"); b = ctx.read_seqdb(f"{c}/aaB")
    out = ctx.concatdbs(a, b)
    out.write(tmp_path / "out")
    assert_same_db(f"{c}/aaC", tmp_path / "out", "concatdbs on thread-ordered inputs")
    # a DB this library wrote is in key order: concatenating the rewritten copies numbers B by key — equal to the reference only where
    # file order and key order agree, i.e. the distinction is real
    b.write(tmp_path / "b_canon"); b2 = ctx.read_seqdb(tmp_path / "b_canon")
    out2 = ctx.concatdbs(a, b2); out2.write(tmp_path / "out2")
    from conftest import read_db
    assert read_db(tmp_path / "out2")[1] != read_db(f"{c}/aaC")[1]
    for x in (a, b, b2, out, out2):
        x.free()
    hip = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "plass_amd", "plass-hip")
    p = subprocess.run([hip, "concatdbs", f"{c}/aaA", f"{c}/aaB", str(tmp_path / "cli")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert p.returncode == 0, p.stdout
    assert_same_db(f"{c}/aaC", tmp_path / "cli", "plass-hip concatdbs on thread-ordered inputs")


def test_concatdbs_header_db_follows_the_file_order_too(ctx, golden, tmp_path):
    """the header DB of a concatenation is renumbered by the same rule as its sequence DB (rank in B's data file,
    DBConcat.cpp:46-47,113-118) — plasship_orfhdr_read keeps the rank, plasship_orfhdr_concat applies it.  Reference-written header
    DBs of two extractorfs passes (key order) and one with a shuffled data file, each against the reference's own concatdbs;
    through the C-ABI and through the command line."""
    import subprocess
    c = os.path.join(golden, "concat")
    hip = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "plass_amd", "plass-hip")
    for bname, cname in (("B_h", "C_h"), ("Bs_h", "Cs_h")):
        a = ctx.read_orfhdr(f"{c}/A_h"); b = ctx.read_orfhdr(f"{c}/{bname}")
        out = ctx.concatdbs(a, b)
        out.write(tmp_path / ("o_" + cname))
        assert_same_db(f"{c}/{cname}", tmp_path / ("o_" + cname), f"header concatdbs {bname}")
        p = subprocess.run([hip, "concatdbs", f"{c}/A_h", f"{c}/{bname}", str(tmp_path / ("cli_" + cname))], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert p.returncode == 0, p.stdout
        assert_same_db(f"{c}/{cname}", tmp_path / ("cli_" + cname), f"plass-hip concatdbs {bname}")
    from conftest import read_db
    assert read_db(f"{c}/C_h")[1] != read_db(f"{c}/Cs_h")[1]            # the order of the file matters
    # the nucleotide ORFs of the same passes
    a = ctx.read_seqdb(f"{c}/A"); b = ctx.read_seqdb(f"{c}/B")
    out = ctx.concatdbs(a, b); out.write(tmp_path / "n")
    assert_same_db(f"{c}/C", tmp_path / "n", "concatdbs of the nucleotide ORF DBs")


def test_concatdbs_preserve_keys(ctx, golden, tmp_path):
    """`concatdbs --preserve-keys` (DBConcat.cpp:113-118; data/nuclassemble.sh:41,145): circular contigs rejoin the linear ones under their own
    keys.  Inputs are index subsets over larger data files (the workflow's `_noneCycle` DB); reference-written DBs
    (tests/golden/make_concat_preserve.sh), through the C-ABI and through `plass-hip concatdbs ... --preserve-keys` as the script calls it"""
    import subprocess
    c = os.path.join(golden, "concat_preserve")
    a, b, rest = ctx.read_seqdb(f"{c}/cycA"), ctx.read_seqdb(f"{c}/cycB"), ctx.read_seqdb(f"{c}/noneCycle")
    call = ctx.concatdbs(a, b, preserve_keys=True); call.write(tmp_path / "cycle_all")
    assert_same_db(f"{c}/cycle_all", tmp_path / "cycle_all", "concatdbs cycA cycB --preserve-keys")
    merged = ctx.concatdbs(rest, call, preserve_keys=True); merged.write(tmp_path / "merged")
    assert_same_db(f"{c}/merged", tmp_path / "merged", "concatdbs noneCycle cycle_all --preserve-keys")
    # the order of the arguments does not matter for a union
    ctx.concatdbs(call, rest, preserve_keys=True).write(tmp_path / "merged2")
    assert_same_db(f"{c}/merged", tmp_path / "merged2", "concatdbs cycle_all noneCycle --preserve-keys")
    # a key held by both DBs is refused (the reference writes two entries under one key)
    with pytest.raises(Exception):
        ctx.concatdbs(call, a, preserve_keys=True)
    hip = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "plass_amd", "plass-hip")
    p = subprocess.run([hip, "concatdbs", f"{c}/noneCycle", f"{c}/cycle_all", str(tmp_path / "cli"), "--preserve-keys"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert p.returncode == 0, p.stdout
    assert_same_db(f"{c}/merged", tmp_path / "cli", "plass-hip concatdbs --preserve-keys")

