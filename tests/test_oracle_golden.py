"""The CPU oracle against everything the reference wrote for this path (golden vectors) — runs without a GPU.
Golden DBs: tests/golden/*.tar.gz, produced by the unmodified reference binaries (tests/golden/make_golden.sh)."""
import ctypes as C
import os

import pytest

from conftest import (AA_AS, AA_KM, AA_RS, GD_AS, GD_KM, GD_P2N, GD_RS, NUCL_AS, NUCL_KM, NUCL_RS, ROOT, aa_iter_flags, assert_same_db, run_oracle,
                      sweep_positional, sweep_variants)


@pytest.mark.parametrize("it", [0, 1, 2])
def test_oracle_aa_iteration(oracle_bin, golden, tmp_path, it):
    s = os.path.join(golden, "aa")
    run_oracle(oracle_bin, ["kmermatcher", f"{s}/seq_{it}", tmp_path / "pref"] + AA_KM + aa_iter_flags(it))
    assert_same_db(f"{s}/pref_{it}", tmp_path / "pref", "kmermatcher")
    run_oracle(oracle_bin, ["rescorediagonal", f"{s}/seq_{it}", f"{s}/seq_{it}", f"{s}/pref_{it}", tmp_path / "aln"] + AA_RS)
    assert_same_db(f"{s}/aln_{it}", tmp_path / "aln", "rescorediagonal")
    run_oracle(oracle_bin, ["assembleresults", f"{s}/seq_{it}", f"{s}/aln_{it}", tmp_path / "seq"] + AA_AS)
    assert_same_db(f"{s}/seq_{it + 1}", tmp_path / "seq", "assembleresults")


def test_oracle_aa_variants(oracle_bin, golden, tmp_path):
    s = os.path.join(golden, "aa")
    run_oracle(oracle_bin, ["rescorediagonal", f"{s}/seq_0", f"{s}/seq_0", f"{s}/pref_0", tmp_path / "aln", "--rescore-mode", "2", "-e", "1e-05",
                            "-c", "0", "-a", "1", "--cov-mode", "0", "--min-seq-id", "0.9"])
    assert_same_db(f"{s}/aln_0_mode2_bt", tmp_path / "aln", "rescore mode 2 + backtrace")
    run_oracle(oracle_bin, ["assembleresults", f"{s}/seq_0", f"{s}/aln_0", tmp_path / "seq", "--min-seq-id", "0.9", "--max-seq-len", "65535",
                            "--keep-target", "0", "--rescore-mode", "3"])
    assert_same_db(f"{s}/seq_1_keeptarget0", tmp_path / "seq", "keep-target 0")


@pytest.mark.parametrize("it", [0, 1])
def test_oracle_nucl_iteration(oracle_bin, golden, tmp_path, it):
    s = os.path.join(golden, "nucl")
    run_oracle(oracle_bin, ["kmermatcher", f"{s}/seq_{it}", tmp_path / "pref"] + NUCL_KM)
    assert_same_db(f"{s}/pref_{it}", tmp_path / "pref", "nucl kmermatcher")
    run_oracle(oracle_bin, ["rescorediagonal", f"{s}/seq_{it}", f"{s}/seq_{it}", f"{s}/pref_{it}", tmp_path / "aln"] + NUCL_RS)
    assert_same_db(f"{s}/aln_{it}", tmp_path / "aln", "nucl rescorediagonal")
    run_oracle(oracle_bin, ["nuclassembleresults", f"{s}/seq_{it}", f"{s}/aln_{it}", tmp_path / "seq"] + NUCL_AS)
    assert_same_db(f"{s}/seq_{it + 1}", tmp_path / "seq", "nuclassembleresults")


def test_oracle_nucleotide_strand_ties(oracle_bin, golden, tmp_path):
    """sort #2 compares (rep, target, diagonal) only (kmermatcher.h:98-130) and the strand of a candidate pair is that of the LAST
    record of the best diagonal's run (kmermatcher.cpp:866-893): a DB on which a (rep, target, diagonal) triple holds records of both
    strands, and what the unmodified reference wrote for it (identical at 1 and 8 threads; tests/golden/make_strand_ties.py).  Rounds
    1-3's rule (reverse strand first) gets this pref entry wrong; `--oracle-old-strand-ties 1` still shows it."""
    s = os.path.join(golden, "strand_ties")
    run_oracle(oracle_bin, ["kmermatcher", f"{s}/seq_0", tmp_path / "pref"] + NUCL_KM)
    assert_same_db(f"{s}/pref_0", tmp_path / "pref", "strand ties kmermatcher")
    run_oracle(oracle_bin, ["rescorediagonal", f"{s}/seq_0", f"{s}/seq_0", f"{s}/pref_0", tmp_path / "aln"] + NUCL_RS)
    assert_same_db(f"{s}/aln_0", tmp_path / "aln", "strand ties rescorediagonal")
    run_oracle(oracle_bin, ["kmermatcher", f"{s}/seq_0", tmp_path / "pref_old"] + NUCL_KM + ["--oracle-old-strand-ties", "1"])
    from conftest import read_db
    assert read_db(tmp_path / "pref_old")[1] != read_db(f"{s}/pref_0")[1]


@pytest.mark.parametrize("it", [0, 1])
def test_oracle_long_nucleotide_contigs(oracle_bin, golden, tmp_path, it):
    """contigs of 17-36 kb that grow to 70 kb: KmerPosition<int> records and the +-65 536 diagonal wrap-around of the 16-bit
    prefilter diagonal (kmermatcher.cpp:797-802, rescorediagonal.cpp:218-240), reverse-strand overlaps between contigs"""
    s = os.path.join(golden, "longnucl")
    run_oracle(oracle_bin, ["kmermatcher", f"{s}/seq_{it}", tmp_path / "pref"] + NUCL_KM)
    assert_same_db(f"{s}/pref_{it}", tmp_path / "pref", "long nucl kmermatcher")
    run_oracle(oracle_bin, ["rescorediagonal", f"{s}/seq_{it}", f"{s}/seq_{it}", f"{s}/pref_{it}", tmp_path / "aln"] + NUCL_RS)
    assert_same_db(f"{s}/aln_{it}", tmp_path / "aln", "long nucl rescorediagonal")
    run_oracle(oracle_bin, ["nuclassembleresults", f"{s}/seq_{it}", f"{s}/aln_{it}", tmp_path / "seq"] + NUCL_AS)
    assert_same_db(f"{s}/seq_{it + 1}", tmp_path / "seq", "long nucl nuclassembleresults")


@pytest.mark.parametrize("ext", [0, 1])
def test_oracle_adversarial_inputs(oracle_bin, golden, tmp_path, ext):
    """hostile inputs (shorter than k, homopolymers and repeats, X / '*' / N / IUPAC / lower case, duplicates and reverse
    complement duplicates, a 33 000-residue contig, index order != key order): what the reference wrote (adversarial.tar.gz)"""
    s = os.path.join(golden, "adversarial")
    e = ["--include-only-extendable", str(ext)]
    run_oracle(oracle_bin, ["kmermatcher", f"{s}/aa_seq", tmp_path / "p"] + AA_KM + ["--hash-shift", "67"] + e)
    assert_same_db(f"{s}/aa_pref{ext}", tmp_path / "p", "adversarial aa kmermatcher")
    run_oracle(oracle_bin, ["rescorediagonal", f"{s}/aa_seq", f"{s}/aa_seq", f"{s}/aa_pref{ext}", tmp_path / "a"] + AA_RS)
    assert_same_db(f"{s}/aa_aln{ext}", tmp_path / "a", "adversarial aa rescorediagonal")
    run_oracle(oracle_bin, ["assembleresults", f"{s}/aa_seq", f"{s}/aa_aln{ext}", tmp_path / "o"] + AA_AS)
    assert_same_db(f"{s}/aa_out{ext}", tmp_path / "o", "adversarial aa assembleresults")
    nkm = [x for x in NUCL_KM]
    nkm[nkm.index("--include-only-extendable") + 1] = str(ext)
    run_oracle(oracle_bin, ["kmermatcher", f"{s}/nucl_seq", tmp_path / "np"] + nkm)
    assert_same_db(f"{s}/n_pref{ext}", tmp_path / "np", "adversarial nucl kmermatcher")
    run_oracle(oracle_bin, ["rescorediagonal", f"{s}/nucl_seq", f"{s}/nucl_seq", f"{s}/n_pref{ext}", tmp_path / "na"] + NUCL_RS)
    assert_same_db(f"{s}/n_aln{ext}", tmp_path / "na", "adversarial nucl rescorediagonal")
    run_oracle(oracle_bin, ["nuclassembleresults", f"{s}/nucl_seq", f"{s}/n_aln{ext}", tmp_path / "no"] + NUCL_AS)
    assert_same_db(f"{s}/n_out{ext}", tmp_path / "no", "adversarial nucl nuclassembleresults")


def test_oracle_guided_iterations(oracle_bin, golden, tmp_path):
    """penguin's protein-guided stage: kmermatcher + rescorediagonal (-a 1) on the protein twins, proteinaln2nucl,
    guidedassembleresults; iteration 0 module by module, iteration 1 chained on the oracle's own DBs"""
    s = os.path.join(golden, "guided")
    t = tmp_path
    run_oracle(oracle_bin, ["kmermatcher", f"{s}/aa_0", t / "pref"] + GD_KM)
    assert_same_db(f"{s}/pref_0", t / "pref", "guided kmermatcher")
    run_oracle(oracle_bin, ["rescorediagonal", f"{s}/aa_0", f"{s}/aa_0", f"{s}/pref_0", t / "aln"] + GD_RS)
    assert_same_db(f"{s}/aln_0", t / "aln", "guided rescorediagonal -a 1")
    run_oracle(oracle_bin, ["proteinaln2nucl", f"{s}/nucl_0", f"{s}/nucl_0", f"{s}/aa_0", f"{s}/aa_0", f"{s}/aln_0", t / "aln_nucl"] + GD_P2N)
    assert_same_db(f"{s}/aln_nucl_0", t / "aln_nucl", "proteinaln2nucl")
    run_oracle(oracle_bin, ["guidedassembleresults", f"{s}/nucl_0", f"{s}/aa_0", f"{s}/aln_nucl_0", t / "nucl_1", t / "aa_1"] + GD_AS)
    assert_same_db(f"{s}/nucl_1", t / "nucl_1", "guidedassembleresults nucl")
    assert_same_db(f"{s}/aa_1", t / "aa_1", "guidedassembleresults aa")
    run_oracle(oracle_bin, ["kmermatcher", t / "aa_1", t / "pref1"] + GD_KM)
    run_oracle(oracle_bin, ["rescorediagonal", t / "aa_1", t / "aa_1", t / "pref1", t / "aln1"] + GD_RS)
    run_oracle(oracle_bin, ["proteinaln2nucl", t / "nucl_1", t / "nucl_1", t / "aa_1", t / "aa_1", t / "aln1", t / "aln_nucl1"] + GD_P2N)
    run_oracle(oracle_bin, ["guidedassembleresults", t / "nucl_1", t / "aa_1", t / "aln_nucl1", t / "nucl_2", t / "aa_2"] + GD_AS)
    assert_same_db(f"{s}/nucl_2", t / "nucl_2", "guided iteration 1 nucl")
    assert_same_db(f"{s}/aa_2", t / "aa_2", "guided iteration 1 aa")


@pytest.mark.parametrize("name,mod,flags", sweep_variants(), ids=[v[0] for v in sweep_variants()])
def test_oracle_flag_sweep(oracle_bin, golden, tmp_path, name, mod, flags):
    """non-default flags (alphabet 21, k, k-mers per sequence, repeated k-mers, the three coverage modes, E-value / identity
    / length thresholds, identity modes, self matches + backtrace, length cap, --keep-target 0): the reference's output"""
    m, pos, outs = sweep_positional(golden, mod, tmp_path / "out")
    run_oracle(oracle_bin, [m] + pos + flags)
    for suffix, path in outs:
        assert_same_db(os.path.join(golden, "sweep", name + suffix), path, name + suffix)


def test_oracle_known_answers(oracle_bin):
    """constants captured from the reference (oracle/ref_tables.h): XXH64, revComplement, bit scores, E-values"""
    lib = C.CDLL(os.path.join(ROOT, "oracle", "build", "liboracle.so"))
    lib.oracle_xxh64_u64.restype = C.c_ulonglong; lib.oracle_xxh64_u64.argtypes = [C.c_ulonglong, C.c_ulonglong]
    lib.oracle_revcomp.restype = C.c_ulonglong; lib.oracle_revcomp.argtypes = [C.c_ulonglong, C.c_int]
    for f in ("oracle_bitscore", "oracle_raw_from_bit"):
        getattr(lib, f).restype = C.c_double; getattr(lib, f).argtypes = [C.c_int, C.c_double]
    lib.oracle_evalue.restype = C.c_double; lib.oracle_evalue.argtypes = [C.c_int, C.c_ulonglong, C.c_double, C.c_double]
    lib.oracle_kat_xxh64.restype = C.POINTER(C.c_ulonglong); lib.oracle_kat_revcomp.restype = C.POINTER(C.c_ulonglong)
    lib.oracle_kat_aa.restype = C.POINTER(C.c_double); lib.oracle_kat_nuc.restype = C.POINTER(C.c_double)
    kx = lib.oracle_kat_xxh64()
    for i in range(4):
        assert lib.oracle_xxh64_u64(kx[3 * i], kx[3 * i + 1]) == kx[3 * i + 2]
    assert lib.oracle_xxh64_u64(12345, 67) == 11599637584503786452
    kr = lib.oracle_kat_revcomp()
    for i in range(3):
        assert lib.oracle_revcomp(kr[3 * i], int(kr[3 * i + 1])) == kr[3 * i + 2]
    ka, kn = lib.oracle_kat_aa(), lib.oracle_kat_nuc()
    assert lib.oracle_bitscore(0, 100.0) == ka[0] and lib.oracle_raw_from_bit(0, 100.0) == ka[1]
    assert lib.oracle_bitscore(1, 100.0) == kn[0] and lib.oracle_raw_from_bit(1, 100.0) == kn[1]
    # E-values agree with the reference to the printed precision (%.3E) and far beyond
    for got, want in ((lib.oracle_evalue(0, 1000000, 60, 50), ka[2]), (lib.oracle_evalue(0, 1000000, 37, 48), ka[3]),
                      (lib.oracle_evalue(0, 1000000, 250, 4000), ka[4]), (lib.oracle_evalue(1, 1000000, 60, 150), kn[2]),
                      (lib.oracle_evalue(1, 1000000, 100, 150), kn[3])):
        assert abs(got - want) <= 1e-12 * abs(want)


@pytest.mark.parametrize("case", [1, 2, 3, 4])
def test_oracle_stale_scan_quirk(oracle_bin, golden, tmp_path, case):
    d = os.path.join(golden, "q1", f"case{case}")
    ext = open(os.path.join(d, "ext")).read().strip()
    flags = AA_KM + ["--hash-shift", "67", "--include-only-extendable", ext]
    run_oracle(oracle_bin, ["kmermatcher", f"{d}/seq", tmp_path / "pref"] + flags)
    assert_same_db(f"{d}/pref", tmp_path / "pref", "stale scan (reference behaviour)")
    run_oracle(oracle_bin, ["kmermatcher", f"{d}/seq", tmp_path / "pref2"] + flags + ["--oracle-no-stale-scan", "1"])
    with pytest.raises(AssertionError):          # the fixture really exercises the quirk
        assert_same_db(f"{d}/pref", tmp_path / "pref2", "without the stale scan")


def test_oracle_findassemblystart(oracle_bin, golden, tmp_path):
    """row N3: the consensus "*M" start detection of iteration 0 (77 of 7 255 example fragments are cut; 14 of the
    translated ORFs of the guided example), and the iteration-0 chain of data/assemble.sh on the corrected sequences"""
    s, g, f = os.path.join(golden, "aa"), os.path.join(golden, "guided"), os.path.join(golden, "fs")
    run_oracle(oracle_bin, ["findassemblystart", f"{s}/seq_0", f"{s}/aln_0", tmp_path / "corr"])
    assert_same_db(f"{f}/corrected_seqs", tmp_path / "corr", "findassemblystart")
    run_oracle(oracle_bin, ["findassemblystart", f"{g}/aa_0", f"{g}/aln_0", tmp_path / "gcorr"])
    assert_same_db(f"{f}/guided_corrected_seqs", tmp_path / "gcorr", "findassemblystart on ORFs")
    run_oracle(oracle_bin, ["kmermatcher", tmp_path / "corr", tmp_path / "pref"] + AA_KM + aa_iter_flags(0))
    run_oracle(oracle_bin, ["rescorediagonal", tmp_path / "corr", tmp_path / "corr", tmp_path / "pref", tmp_path / "aln"] + AA_RS)
    run_oracle(oracle_bin, ["assembleresults", tmp_path / "corr", tmp_path / "aln", tmp_path / "as0"] + AA_AS)
    assert_same_db(f"{f}/assembly_0", tmp_path / "as0", "iteration 0 on the corrected sequences")


def test_oracle_cyclecheck(oracle_bin, golden, tmp_path):
    """row N4: circular genomes assembled past their end (exact / 2 % errors, 300 nt - 47 kb), linear controls, repeats, N runs,
    lower case, tiny sequences, the --max-seq-len boundary, reads; plus the example's contigs (none circular)"""
    c = os.path.join(golden, "cyc")
    for chop in (0, 1):
        run_oracle(oracle_bin, ["cyclecheck", f"{c}/in", tmp_path / f"o{chop}", "--max-seq-len", "50000", "--chop-cycle", chop])
        assert_same_db(f"{c}/cycle_chop{chop}", tmp_path / f"o{chop}", f"cyclecheck --chop-cycle {chop}")
    for src, name in ((os.path.join(golden, "nucl", "seq_2"), "nucl_seq_2"), (os.path.join(golden, "longnucl", "seq_0"), "longnucl_seq_0"),
                      (os.path.join(golden, "longnucl", "seq_2"), "longnucl_seq_2")):
        run_oracle(oracle_bin, ["cyclecheck", src, tmp_path / name, "--max-seq-len", "200000", "--chop-cycle", "1"])
        assert_same_db(f"{c}/{name}_cycle", tmp_path / name, f"cyclecheck {name}")


@pytest.mark.parametrize("threads", [3, 8])
def test_oracle_thread_count_independent(oracle_bin, golden, tmp_path, threads):
    """--threads (OpenMP over extraction, the sorts, re-scoring, extension): the golden DBs whatever the thread count"""
    s = os.path.join(golden, "aa")
    t = ["--threads", threads]
    run_oracle(oracle_bin, ["kmermatcher", f"{s}/seq_1", tmp_path / "pref"] + AA_KM + aa_iter_flags(1) + t)
    assert_same_db(f"{s}/pref_1", tmp_path / "pref", "kmermatcher")
    run_oracle(oracle_bin, ["rescorediagonal", f"{s}/seq_1", f"{s}/seq_1", f"{s}/pref_1", tmp_path / "aln"] + AA_RS + t)
    assert_same_db(f"{s}/aln_1", tmp_path / "aln", "rescorediagonal")
    run_oracle(oracle_bin, ["assembleresults", f"{s}/seq_1", f"{s}/aln_1", tmp_path / "seq"] + AA_AS + t)
    assert_same_db(f"{s}/seq_2", tmp_path / "seq", "assembleresults")
    n = os.path.join(golden, "nucl")
    run_oracle(oracle_bin, ["kmermatcher", f"{n}/seq_1", tmp_path / "npref"] + NUCL_KM + t)
    assert_same_db(f"{n}/pref_1", tmp_path / "npref", "nucl kmermatcher")
    run_oracle(oracle_bin, ["rescorediagonal", f"{n}/seq_1", f"{n}/seq_1", f"{n}/pref_1", tmp_path / "naln"] + NUCL_RS + t)
    assert_same_db(f"{n}/aln_1", tmp_path / "naln", "nucl rescorediagonal")
    run_oracle(oracle_bin, ["nuclassembleresults", f"{n}/seq_1", f"{n}/aln_1", tmp_path / "nseq"] + NUCL_AS + t)
    assert_same_db(f"{n}/seq_2", tmp_path / "nseq", "nuclassembleresults")


def test_oracle_orf_preprocessing(oracle_bin, golden, tmp_path):
    """row N2 (oracle only so far): extractorfs + translatenucs on hostile reads for four flag sets (sequences, headers,
    translations with and without --add-orf-stop), and the example's whole preprocessing chain — read DB -> two extractorfs
    passes -> translatenucs -> concatdbs — which must reproduce the protein fragment DB iteration 0 starts from (aa/seq_0)"""
    o = os.path.join(golden, "orf")
    flags = [l.split() for l in open(os.path.join(o, "FLAGS")).read().splitlines() if l.strip()]
    for n, fl in zip((1, 2, 4, 5), flags):
        run_oracle(oracle_bin, ["extractorfs", f"{o}/in", tmp_path / f"orfs_{n}"] + fl)
        assert_same_db(f"{o}/orfs_{n}", tmp_path / f"orfs_{n}", f"extractorfs flag set {n}")
        assert_same_db(f"{o}/orfs_{n}_h", str(tmp_path / f"orfs_{n}") + "_h", f"extractorfs headers, flag set {n}")
        run_oracle(oracle_bin, ["translatenucs", f"{o}/orfs_{n}", tmp_path / f"aa_stop_{n}", "--translation-table", "1", "--add-orf-stop", "1"])
        assert_same_db(f"{o}/aa_stop_{n}", tmp_path / f"aa_stop_{n}", f"translatenucs --add-orf-stop 1, flag set {n}")
        run_oracle(oracle_bin, ["translatenucs", f"{o}/orfs_{n}", tmp_path / f"aa_{n}", "--translation-table", "1", "--add-orf-stop", "0"])
        assert_same_db(f"{o}/aa_{n}", tmp_path / f"aa_{n}", f"translatenucs --add-orf-stop 0, flag set {n}")
    small = [l.split() for l in open(os.path.join(o, "FLAGS_SMALL")).read().splitlines() if l.strip()]     # any-to-stop; --translate 1
    for name, fl in zip(("small_orfs_any", "small_orfs_translated"), small):
        run_oracle(oracle_bin, ["extractorfs", f"{o}/in_small", tmp_path / name] + fl)
        assert_same_db(f"{o}/{name}", tmp_path / name, f"extractorfs {name}")
        assert_same_db(f"{o}/{name}_h", str(tmp_path / name) + "_h", f"extractorfs {name} headers")
    reads = os.path.join(golden, "nucl", "seq_0")              # nucl_reads of the bundled example (mergereads output)
    common = ["--max-gaps", "0", "--orf-start-mode", "0", "--forward-frames", "1,2,3", "--reverse-frames", "1,2,3", "--translation-table", "1",
              "--translate", "0", "--use-all-table-starts", "0"]
    run_oracle(oracle_bin, ["extractorfs", reads, tmp_path / "nucl_6f_start", "--min-length", "20", "--max-length", "45", "--contig-start-mode", "1",
                            "--contig-end-mode", "0"] + common)
    run_oracle(oracle_bin, ["extractorfs", reads, tmp_path / "nucl_6f_long", "--min-length", "45", "--max-length", "32734", "--contig-start-mode", "2",
                            "--contig-end-mode", "2"] + common)
    for n in ("start", "long"):
        run_oracle(oracle_bin, ["translatenucs", tmp_path / f"nucl_6f_{n}", tmp_path / f"aa_6f_{n}", "--translation-table", "1", "--add-orf-stop", "1"])
    run_oracle(oracle_bin, ["concatdbs", tmp_path / "aa_6f_long", tmp_path / "aa_6f_start", tmp_path / "aa_6f_start_long"])
    assert_same_db(os.path.join(golden, "aa", "seq_0"), tmp_path / "aa_6f_start_long", "preprocessing chain -> aa_6f_start_long")


def test_oracle_concatdbs_follows_the_data_file_order(oracle_bin, golden, tmp_path):
    """concatdbs numbers its SECOND DB in data-file order (DBConcat.cpp:46-47,113-118): inputs the reference's translatenucs wrote
    on 8 threads (data in thread order, raw files — not canonicalised) and the reference's own concatenation"""
    c = os.path.join(golden, "concat")
    idx = [tuple(int(x) for x in l.split()[:3]) for l in open(os.path.join(c, "aaB.index"))]
    assert any(a[1] > b[1] for a, b in zip(idx, idx[1:])), "the fixture must hold a DB whose file order is not its key order"
    run_oracle(oracle_bin, ["concatdbs", os.path.join(c, "aaA"), os.path.join(c, "aaB"), tmp_path / "out"])
    assert_same_db(os.path.join(c, "aaC"), tmp_path / "out", "concatdbs on thread-ordered inputs")
    # the header DBs of the same passes (key order as extractorfs leaves them), and one whose data file is shuffled
    run_oracle(oracle_bin, ["concatdbs", os.path.join(c, "A_h"), os.path.join(c, "B_h"), tmp_path / "h"])
    assert_same_db(os.path.join(c, "C_h"), tmp_path / "h", "concatdbs of the header DBs")
    run_oracle(oracle_bin, ["concatdbs", os.path.join(c, "A_h"), os.path.join(c, "Bs_h"), tmp_path / "hs"])
    assert_same_db(os.path.join(c, "Cs_h"), tmp_path / "hs", "concatdbs of a header DB in shuffled file order")
    run_oracle(oracle_bin, ["concatdbs", os.path.join(c, "A"), os.path.join(c, "B"), tmp_path / "n"])
    assert_same_db(os.path.join(c, "C"), tmp_path / "n", "concatdbs of the nucleotide ORF DBs")


def test_oracle_concatdbs_preserve_keys(oracle_bin, golden, tmp_path):
    """`concatdbs --preserve-keys` as data/nuclassemble.sh:41,145 calls it (DBConcat.cpp:113-118 with preserveKeysB): the union of two DBs with
    disjoint keys, inputs that are index subsets over a larger data file like the workflow's `_noneCycle` DB; reference-written
    (tests/golden/make_concat_preserve.sh)"""
    c = os.path.join(golden, "concat_preserve")
    run_oracle(oracle_bin, ["concatdbs", os.path.join(c, "cycA"), os.path.join(c, "cycB"), tmp_path / "cycle_all", "--preserve-keys", "1"])
    assert_same_db(os.path.join(c, "cycle_all"), tmp_path / "cycle_all", "concatdbs cycA cycB --preserve-keys")
    run_oracle(oracle_bin, ["concatdbs", os.path.join(c, "noneCycle"), tmp_path / "cycle_all", tmp_path / "merged", "--preserve-keys", "1"])
    assert_same_db(os.path.join(c, "merged"), tmp_path / "merged", "concatdbs noneCycle cycle_all --preserve-keys")


def test_oracle_strand_ties_are_a_reference_outcome(oracle_bin, tmp_path):
    """where the reference's nucleotide kmermatcher is not deterministic (strand-tied pairs: 10 runs of one command gave 8
    versions of one entry), the oracle's result must be ONE OF THE REFERENCE'S — checked as set membership, line by line, against
    tests/golden/strand_membership.json (the judge's case: seed 424242, 40 000 pairs, 4 genomes of 100-200 kb, three iterations of the
    nucleotide chain; 10 reference runs per iteration; generator tests/golden/make_strand_membership.py)."""
    import json
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from make_strand_membership import chain_step, entries
    from conftest import check_strand_membership
    import conftest as T
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "strand_membership.json")))
    assert sum(r["tie_dependent_pairs"] for r in fx["iterations"]) >= 10 and "UNMODIFIED reference" in fx["made_by"]
    P = lambda n: str(tmp_path / n)
    run_oracle(oracle_bin, ["synthreads", P("reads")] + fx["synth"])
    src = P("reads")
    for it, rec in enumerate(fx["iterations"]):
        run_oracle(oracle_bin, ["kmermatcher", src, P("o_pref")] + fx["kmermatcher"] + ["--threads", "4"])
        ent = entries(P("o_pref"))
        check_strand_membership(rec, ent, "oracle, nucleotide iteration %d" % it)
        for k, t in rec["ties"].items():                      # (the oracle is deterministic: it writes what it wrote when the fixture was made)
            assert ent[int(k)].decode("latin-1") == t["oracle"]
        src = chain_step(oracle_bin, T, src, P, it, 4)


def test_oracle_circular_chain(oracle_bin, tmp_path):
    """`cyclecheck --chop-cycle 1` AT DEPTH.  A community with small circular replicons (tests/golden/make_circular_chain.py:
    numpy reads that wrap around the origin, seed in the fixture); contigs grow around the replicons until their ends overlap and cyclecheck takes
    them out in iterations 3-6 (9 / 94 / 167 / 233 contigs).  Every DB of the seven iterations equals tests/golden/circular_chain.json, whose
    every module output was compared with the unmodified reference's on the same inputs (28 of 28: profiles/r06_circular_chain_pin.txt)."""
    import json
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from make_circular_chain import write_reads
    from make_large_nucl import db_sums, rest_db
    import conftest as T
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "circular_chain.json")))
    assert "unmodified reference" in fx["made_by"] and sum(r["n_cyclic"] for r in fx["iterations"][3:]) > 100
    P = lambda n: str(tmp_path / n)
    write_reads(P("reads"))
    assert db_sums(P("reads")) == fx["reads"]
    src = P("reads")
    km = T.NUCL_KM + ["--max-seq-len", "200000"]
    for it, want in enumerate(fx["iterations"]):
        run_oracle(oracle_bin, ["kmermatcher", src, P("pref")] + km + ["--threads", "4"])
        run_oracle(oracle_bin, ["rescorediagonal", src, src, P("pref"), P("aln")] + T.NUCL_RS + ["--threads", "4"])
        run_oracle(oracle_bin, ["nuclassembleresults", src, P("aln"), P("assembly_%d" % it)] + T.NUCL_AS + ["--threads", "4"])
        run_oracle(oracle_bin, ["cyclecheck", P("assembly_%d" % it), P("cycle_%d" % it), "--max-seq-len", "200000", "--chop-cycle", "1"])
        assert rest_db(P("assembly_%d" % it), P("cycle_%d" % it), P("rest_%d" % it)) == want["n_cyclic"]
        for name, path in (("pref", P("pref")), ("aln", P("aln")), ("assembly", P("assembly_%d" % it)), ("cycle", P("cycle_%d" % it)), ("rest", P("rest_%d" % it))):
            assert db_sums(path) == want[name], "iteration %d: %s" % (it, name)
        src = P("rest_%d" % it)

