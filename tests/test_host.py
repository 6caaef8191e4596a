"""Host-side checks that need no GPU: the C-ABI library exports what include/plasship.h declares, fails loudly
without a device, and the synthetic generator is deterministic."""
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT


def _declared_symbols():
    """every function the C-ABI headers declare (include/plasship.h, plasship_synth.h, plasship_rccl.h)"""
    names = set()
    for h in sorted(os.listdir(os.path.join(ROOT, "include"))):
        if h.endswith(".h"):
            names |= set(re.findall(r"\b(plasship_[a-z_0-9]+)\s*\(", open(os.path.join(ROOT, "include", h)).read()))
    return sorted(names)


def test_library_exports_every_declared_symbol():
    import plass_amd
    from plass_amd import _lib
    if not os.path.exists(plass_amd.lib_path()):
        import __graft_entry__ as g
        g.build()
    lib = plass_amd.load_library()
    declared = _declared_symbols()
    assert len(declared) >= 24
    bound = {s[0] for s in _lib.SYMBOLS + _lib.SYNTH_SYMBOLS + _lib.RCCL_SYMBOLS}
    for name in declared:
        assert hasattr(lib, name), "missing export " + name
        assert name in bound, "python binding does not cover " + name
    assert lib.plasship_version().startswith(b"plasship")


def test_host_boundary_files(tmp_path):
    """the threaded DB reader / writers of the product (plass_amd/csrc/host_util.cpp) without a GPU: byte-identical files for 1, 3 and
    16 host threads, data split over NAME.0..NAME.2 read back, bad indices refused, failed writes leave nothing behind"""
    import subprocess
    exe = tmp_path / "host_io_check"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", os.path.join(ROOT, "tests", "tools", "host_io_check.cpp"),
                           os.path.join(ROOT, "plass_amd", "csrc", "host_util.cpp"), "-o", str(exe)])
    for threads in ("1", "3", "16"):
        d = tmp_path / ("t" + threads)
        d.mkdir()
        out = subprocess.run([str(exe), str(d)], env=dict(os.environ, PLASSHIP_HOST_THREADS=threads), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert out.returncode == 0 and "host_io_check ok (%s host threads)" % threads in out.stdout, out.stdout


def test_no_cpu_fallback():
    """without a GPU the product must refuse to run (no silent CPU path)"""
    import plass_amd
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(plass_amd.PlasshipError):
        plass_amd.Context(0)


def test_product_does_not_reference_oracle():
    """the product tree never includes, links or executes anything under oracle/"""
    for d, _, files in os.walk(os.path.join(ROOT, "plass_amd")):
        if "build" in d:
            continue
        for f in files:
            if f.endswith((".hip", ".cpp", ".hpp", ".h", ".py", "Makefile")):
                txt = open(os.path.join(d, f), errors="ignore").read()
                for line in txt.splitlines():
                    if re.search(r'#include\s+"[^"]*oracle|^\s*(import|from)\s+oracle\b|oracle/build|liboracle|plass_oracle', line):
                        raise AssertionError("%s references the oracle: %s" % (f, line))


def test_synth_deterministic():
    from plass_amd import synth
    a = synth.protein_fragment_db(300, seed=3)
    b = synth.protein_fragment_db(300, seed=3)
    assert a[0] == b[0] and np.array_equal(a[2], b[2])
    data, off, elen, key = a
    assert len(key) > 300 and (elen - 2).min() >= 45 and (elen - 2).max() <= 50
    for o, l in zip(off[:50], elen[:50]):
        e = data[int(o):int(o) + int(l)]
        assert e.endswith(b"\n\x00") and b"*" not in e[:-3]


def test_bench_configs_and_scaled_workloads():
    """bench.py's workloads: the configs are BASELINE.json's, --pairs keeps the coverage of the community model, and the
    per-iteration hash shift is the workflow's (src/workflow/Assembler.cpp:99-110)"""
    import bench
    assert [bench.hash_shift(i) for i in range(6)] == [67, 68, 68, 69, 69, 70]
    c3 = bench.synth_params("c3")
    assert (c3.n_pairs, c3.n_genomes, c3.genome_min_len, c3.genome_max_len, c3.abundance_sigma, c3.seed) == (25000000, 200, 1000000, 5000000, 1.0, 2)
    c2 = bench.synth_params("c2")
    assert (c2.n_pairs, c2.n_genomes, c2.genome_min_len, c2.seed) == (500000, 1, 7500000, 1)
    cov = lambda p: 300.0 * p.n_pairs / (p.n_genomes * (p.genome_min_len + p.genome_max_len) / 2)
    for pairs in (120000, 1000000, 5000000):
        s = bench.synth_params("c3", pairs)
        assert s.n_genomes >= 1 and 0.5 * cov(c3) < cov(s) < 2.0 * cov(c3)
    s = bench.synth_params("c2", 50000)
    assert abs(cov(s) - cov(c2)) < 0.01 * cov(c2)


def test_local_group_host_allgather_threads():
    """the in-process communicator of the GPU sharding tests (plass_amd.shard.LocalGroup): its host all-gather between
    three rank threads, driven through the C callback table like libplasship drives it (no GPU needed for this part)"""
    import ctypes as C
    import threading
    from plass_amd.shard import LocalGroup, owned_range
    g = LocalGroup(3)
    got = [None] * 3

    def work(r):
        send = (C.c_uint64 * 2)(r, 100 + r); recv = (C.c_uint64 * 6)()
        for _ in range(5):                       # several rounds: the barrier is reusable
            assert g.comms[r].struct.allgather_host(None, C.addressof(send), C.addressof(recv), 16) == 0
        got[r] = list(recv)

    th = [threading.Thread(target=work, args=(r,)) for r in range(3)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=60)
    assert all(x == [0, 100, 1, 101, 2, 102] for x in got)
    # ownership ranges tile [0, n) for any n and world
    for n in (0, 1, 7, 1000, 1622575):
        for w in (1, 2, 3, 8):
            edges = [owned_range(n, r, w) for r in range(w)]
            assert edges[0][0] == 0 and edges[-1][1] == n and all(edges[i][1] == edges[i + 1][0] for i in range(w - 1))


def test_large_fixture_matches_the_generator(tmp_path):
    """tests/golden/large_chain.json (CPU-oracle checksums of the 5 M-read chain) was made for exactly the parameters bench.py
    derives for that size, and the CPU read generator (`plass_oracle synthreads`, the read model shared with the GPU generator) is
    deterministic: same parameters, same digest; another seed, another digest"""
    import json
    import subprocess
    import bench
    import __graft_entry__ as g
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "large_chain.json")))
    sp = bench.synth_params(gold["config"], gold["pairs"])
    for k, v in gold["synth"].items():
        assert getattr(sp, k) == pytest.approx(v), k
    assert gold["reads"]["entries"] == 2 * gold["pairs"] and gold["reads"]["bytes"] == 2 * gold["pairs"] * (sp.read_len + 2)
    assert len(gold["iterations"]) >= 3 and all(it["seq"]["entries"] == gold["fragments"]["entries"] for it in gold["iterations"])
    if not os.path.exists(g.oracle_bin()):
        subprocess.check_call(["make", "-j", "4"], cwd=os.path.join(ROOT, "oracle"))

    def digest(name, seed):
        p = str(tmp_path / name)
        g.run_oracle(["synthreads", p, "--pairs", "3000", "--seed", str(seed), "--genomes", "2", "--genome-min-len", "30000", "--genome-max-len", "60000",
                      "--abundance-sigma", "1.0"])
        out = subprocess.run([g.oracle_bin(), "dbsum", p], stdout=subprocess.PIPE, check=True, text=True).stdout
        return dict(x.split("=") for x in out.strip().split("\t")[1:])

    a, b, c = digest("a", 2), digest("b", 2), digest("c", 3)
    assert a == b and a["entries"] == "6000" and a["digest"] != c["digest"]


def test_rccl_stub_exports_what_the_native_communicator_resolves():
    """tests/tools/rccl_stub.cpp (the in-process librccl stand-in of the multi-rank tests) exports every nccl* symbol
    plass_amd/csrc/comm_rccl.hip looks up with dlsym — so a W > 1 run through it executes the product's exchange() unchanged"""
    import ctypes
    import __graft_entry__ as g
    stub = os.path.join(ROOT, "tests", "tools", "librccl_stub.so")
    if not os.path.exists(stub):
        g.build()
    wanted = set(re.findall(r'sym\("(nccl[A-Za-z]+)"\)', open(os.path.join(ROOT, "plass_amd", "csrc", "comm_rccl.hip")).read()))
    assert len(wanted) == 10
    lib = ctypes.CDLL(stub)
    for name in sorted(wanted):
        assert hasattr(lib, name), "the stub lacks " + name


def test_stored_traffic_is_quoted_only_for_the_sources_it_was_measured_on(tmp_path, monkeypatch):
    """bench.py reads `roofline.traffic` from profiles/<bench.PMC_FILES[config]> (two rocprofv3 --pmc passes of the driver's command) and
    must refuse the file when the product sources differ from the ones recorded in it (a stored figure must not go
    stale silently)."""
    import json
    import bench
    f = next(os.path.join(ROOT, "profiles", n) for n in bench.PMC_FILES["c3"] if os.path.exists(os.path.join(ROOT, "profiles", n)))      # newest stored pass
    rows = json.load(open(f))
    assert rows["steps"] > 0 and rows["kernels"] and "--steps 20 --warmup 5" in rows["source"]
    # for the sources the file was measured on the figure is quoted ...
    assert len(bench.source_sha()) == 16
    monkeypatch.setattr(bench, "source_sha", lambda: rows["source_sha"])
    traffic, note = bench.stored_traffic("extractKernel", 1.0)
    assert traffic and traffic > 1e9 and "rocprofv3" in note
    # ... all instantiations of the kernel template are summed (rocprofv3 lists them as separate symbols)
    # (round 6: the row kernels and their list binning run between the wave tiers' events and belong to the same row: bench.KERNEL_SYMBOLS)
    import re
    per = [r["hbm_bytes_per_launch"] * r["launches"] / rows["steps"] for k, r in rows["kernels"].items() if re.search(bench.KERNEL_SYMBOLS["extractKernel"], k)]
    assert len(per) >= 3 and traffic == pytest.approx(sum(per))
    # ... and any other source hash is refused
    monkeypatch.setattr(bench, "source_sha", lambda: "0" * 16)
    traffic, note = bench.stored_traffic("extractKernel", 1.0)
    assert traffic is None and "other sources" in note


def test_committed_headline_digests_are_complete():
    """tests/golden/c3_chain_digests.json: one digest per output DB of the 12 iterations of the 50 M-read chain (what bench.py's
    `verify` compares a run with), labelled as self-generated (three implementations agreeing: tests/test_gpu_large.py)"""
    import json
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "c3_chain_digests.json")))
    digs = g.get("seq_digests") or g.get("digests")
    assert digs and len(digs) == 12 and len(set(digs)) == 12 and all(len(d) == 16 for d in digs)


def test_window_score_is_the_low_16_bits_of_xxh64(tmp_path):
    """plass_amd/csrc/xxh64_u64.hpp: the extraction kernels take a window's 16-bit score from xxh64Score16, which skips the parts of the
    last 64-bit multiplication the low 16 bits do not depend on — compiled here with g++ against xxh64U64, the oracle's restatement of
    XXH64 and the reference's known answers (oracle.kat_xxh64)"""
    import subprocess, ctypes
    src = tmp_path / "t.cpp"
    src.write_text('#include "xxh64_u64.hpp"\n#include <cstdio>\n#include <random>\n'
                   'extern "C" unsigned long long full(unsigned long long v, unsigned long long s) { return plasship::xxh64U64(v, s); }\n'
                   'extern "C" unsigned score(unsigned long long v, unsigned long long s) { return plasship::xxh64Score16(v, s); }\n'
                   'extern "C" long sweep(long n) { std::mt19937_64 g(7); long bad = 0; for (long i = 0; i < n; i++) { unsigned long long v = g(), s = (i & 3) ? (g() & 0xFF) : g();'
                   ' if (i % 5 == 0) v &= 0xFFFFFFFFFFFFFULL; bad += plasship::xxh64Score16(v, s) != (unsigned) (plasship::xxh64U64(v, s) & 0xFFFF); } return bad; }\n')
    so = tmp_path / "t.so"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-I", os.path.join(ROOT, "plass_amd", "csrc"), str(src), "-o", str(so)])
    lib = ctypes.CDLL(str(so))
    lib.full.restype = ctypes.c_ulonglong; lib.full.argtypes = [ctypes.c_ulonglong, ctypes.c_ulonglong]
    lib.score.restype = ctypes.c_uint; lib.score.argtypes = [ctypes.c_ulonglong, ctypes.c_ulonglong]
    lib.sweep.restype = ctypes.c_long; lib.sweep.argtypes = [ctypes.c_long]
    assert lib.sweep(20000000) == 0
    import __graft_entry__ as g
    g.oracle_bin()                                          # builds oracle/build/ if it is missing
    ora = ctypes.CDLL(os.path.join(ROOT, "oracle", "build", "liboracle.so"))
    ora.oracle_xxh64_u64.restype = ctypes.c_ulonglong; ora.oracle_xxh64_u64.argtypes = [ctypes.c_ulonglong, ctypes.c_ulonglong]
    for v, s in ((0, 0), (1, 67), (0xFFFFFFFFFFFFFFFF, 68), (3899999999999999, 70), (123456789012345, 2 ** 63 + 5)):
        assert lib.full(v, s) == ora.oracle_xxh64_u64(v, s) and lib.score(v, s) == (ora.oracle_xxh64_u64(v, s) & 0xFFFF)
    assert lib.full(12345, 67) == 11599637584503786452


def test_posterior_class_route_agrees_with_the_reference_formula(tmp_path):
    """plass_amd/csrc/posterior_class.hpp: the nucleotide comparator's posterior p from term ratios and an integer lgamma (what the kernels
    evaluate) against the reference's formula — four lgamma, exp and log per term (src/assembler/nuclassembleresult.cpp:36-58, restated
    below) — under the host's libm, on overlap tuples from reads (30 columns) to contigs (260 000 columns): the difference must stay
    well inside the band within which the kernels leave the decision to the host-evaluated table, so outside the band both give one class."""
    import subprocess, ctypes
    src = tmp_path / "p.cpp"
    src.write_text(r'''
#include "posterior_class.hpp"
#include <cstdio>
#include <random>
#include <algorithm>
static double pRef(unsigned alpha1, unsigned beta1, unsigned alpha2, unsigned beta2) {
    const double log_c = (lgamma((double) (beta1 + beta2)) + lgamma((double) (alpha1 + beta1))) - (lgamma((double) (alpha1 + beta1 + beta2)) + lgamma((double) beta1));
    double log_r = 0.0, p = 0.0;
    for (size_t idx = 0; idx < alpha2; idx++) {
        p += exp(log_r + log_c);
        log_r = log((double) (alpha1 + idx)) + log((double) (beta2 + idx)) - (log((double) (idx + 1)) + log((double) (idx + alpha1 + beta1 + beta2))) + log_r;
    }
    return p;
}
extern "C" double sweep(long n, long *classDiff) {
    std::mt19937_64 g(5);
    double worst = 0; *classDiff = 0;
    for (long it = 0; it < n; it++) {
        const unsigned maxL = (it % 4 == 0) ? 260000 : ((it % 4 == 1) ? 20000 : ((it % 4 == 2) ? 600 : 40));
        const unsigned l1 = 1 + g() % maxL, l2 = (g() % 3 == 0) ? 1 + g() % maxL : (unsigned) std::max<long>(1, (long) l1 + (long) (g() % 41) - 20);
        const double e1 = (g() % 1000) / 1000.0 * 0.1, e2 = (g() % 3) ? e1 * (0.5 + (g() % 1000) / 1000.0) : (g() % 1000) / 1000.0 * 0.1;
        const unsigned m1 = std::min<unsigned>(l1, (unsigned) (l1 * e1)), m2 = std::min<unsigned>(l2, (unsigned) (l2 * e2));
        const unsigned a1 = m1 + 1, b1 = l1 - m1 + 1, a2 = m2 + 1, b2 = l2 - m2 + 1;
        const double a = plasship::nuclPosteriorP(a1, b1, a2, b2), b = pRef(a1, b1, a2, b2), band = plasship::nuclPosteriorBand(a1, b1, a2, b2);
        worst = std::max(worst, std::fabs(a - b) / band);
        const bool inBand = std::fabs(a - 0.45) < band || std::fabs(a - 0.55) < band;
        const int ca = a < 0.45 ? 0 : (a > 0.55 ? 1 : 2), cb = b < 0.45 ? 0 : (b > 0.55 ? 1 : 2);
        if (!inBand && ca != cb) (*classDiff)++;
    }
    return worst;
}
''')
    so = tmp_path / "p.so"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-I", os.path.join(ROOT, "plass_amd", "csrc"), str(src), "-o", str(so)])
    lib = ctypes.CDLL(str(so))
    lib.sweep.restype = ctypes.c_double; lib.sweep.argtypes = [ctypes.c_long, ctypes.POINTER(ctypes.c_long)]
    diff = ctypes.c_long(0)
    worst = lib.sweep(60000, ctypes.byref(diff))
    assert diff.value == 0, "a class differs outside the band"
    assert worst < 0.25, "the two routes differ by %.2f of the band" % worst


def test_cli_exit_codes_separate_unsupported_from_failed():
    """plass-hip's exit-code contract (no GPU needed with PLASSHIP_CLI_DRYRUN=1): 95 = the call is well-formed
    for the reference but outside the GPU path (the wrapper hands it to the reference binary: linclust's rescorediagonal at the end of
    `penguin guided_nuclassemble`, lib/mmseqs/data/workflow/linclust.sh:30), 96 = accepted in a dry run, 1 = a malformed call"""
    exe = os.path.join(ROOT, "plass_amd", "plass-hip")
    env = dict(os.environ, PLASSHIP_CLI_DRYRUN="1")
    def rc(*args):
        return subprocess.run([exe] + list(args), env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True).returncode
    assert rc("rescorediagonal", "q", "t", "p", "o", "--rescore-mode", "0", "--wrapped-scoring", "1", "-e", "0.001", "--min-seq-id", "0.9") == 95
    assert rc("rescorediagonal", "q", "t", "p", "o", "--rescore-mode", "3", "--filter-hits", "1") == 95
    assert rc("kmermatcher", "s", "p", "--kmer-per-seq", "21", "--mask", "1", "-k", "14") == 95
    assert rc("kmermatcher", "s", "p", "--kmer-per-seq", "21") == 95                         # automatic k
    assert rc("linclust", "a", "b", "c") == 95 and rc("clust", "a", "b", "c") == 95         # not hot-path modules
    assert rc("rescorediagonal", "q", "t", "p", "o", "--rescore-mode", "3", "-e", "1e-5", "--min-seq-id", "0.9", "-c", "0", "--threads", "4") == 96
    assert rc("kmermatcher", "s", "p", "-k", "22", "--kmer-per-seq", "60", "--alph-size", "nucl:5,aa:13", "--spaced-kmer-mode", "0", "--mask", "0",
              "--sub-mat", "nucl:nucleotide.out,aa:blosum62.out", "--cov-mode", "1", "-c", "0.99", "--min-seq-id", "0.97") == 96
    assert rc("kmermatcher", "s", "p", "-k", "14", "--kmer-per-seq", "60", "--no-such-flag", "1") == 1
    assert rc("concatdbs", "a", "b", "c", "--preserve-keys") == 96                           # implemented since round 6 (sequence DBs)
    # valid for the reference, outside the GPU path: refused with 95 BEFORE the dry-run exit (they used to end with 1 after it)
    assert rc("concatdbs", "a", "b", "c", "--take-larger-entry") == 95
    assert rc("proteinaln2nucl", "qn", "tn", "qa", "ta", "aln", "out") == 95                 # separate query / target DBs
    assert rc("proteinaln2nucl", "n", "n", "a", "a", "aln", "out") == 96
    assert rc("proteinaln2nucl", "n", "n", "qa", "ta", "aln", "out") == 1                    # malformed for the reference as well
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        for name, ty in (("aln", 5), ("hdr", 12), ("seq", 1)):
            open(os.path.join(td, name + ".dbtype"), "wb").write(ty.to_bytes(4, "little"))
        P = lambda n: os.path.join(td, n)
        assert rc("concatdbs", P("aln"), P("aln"), P("o")) == 95                             # an alignment DB: the probe of <A>.dbtype sits before the dry-run exit
        assert rc("concatdbs", P("hdr"), P("hdr"), P("o")) == 96 and rc("concatdbs", P("seq"), P("seq"), P("o"), "--preserve-keys") == 96
        assert rc("concatdbs", P("hdr"), P("hdr"), P("o"), "--preserve-keys") == 95


def test_wrapper_routes_by_exit_code(tmp_path):
    """plass_amd/plass-gpu-wrapper: hot-path modules go to plass-hip, exit 95 and every other module go to the reference binary, which is
    exec'ed under the WRAPPER's name (the reference exports MMSEQS = argv[0], Application.cpp:198)"""
    ref = tmp_path / "ref"
    ref.write_text("#!/bin/bash\necho \"REF argv0=$0 args=$*\"\n")
    ref.chmod(0o755)
    w = os.path.join(ROOT, "plass_amd", "plass-gpu-wrapper")
    log = tmp_path / "log"
    env = dict(os.environ, PLASSHIP_CLI_DRYRUN="1", PLASS_REF_BIN=str(ref), PLASS_WRAPPER_LOG=str(log))
    out = subprocess.run([w, "rescorediagonal", "a", "a", "p", "o", "--rescore-mode", "0", "--wrapped-scoring", "1"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert out.returncode == 0 and "REF argv0=" in out.stdout and "args=rescorediagonal a a p o --rescore-mode 0" in out.stdout
    out = subprocess.run([w, "createdb", "x", "y"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert "args=createdb x y" in out.stdout
    out = subprocess.run([w, "kmermatcher", "s", "p", "-k", "14", "--kmer-per-seq", "60", "--bogus", "1"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert out.returncode == 1 and "REF" not in out.stdout and "Unrecognized parameter" in out.stdout
    lines = log.read_text().splitlines()
    assert lines[0].startswith("reference  <- plass-hip exit 95") and "not a hot-path module" in lines[1] and lines[2].startswith("GPU path   exit 1")
    # NOT a dry run: a request that is valid for the reference and outside the GPU path reaches the reference — plass-hip refuses
    # it before it creates a context, so this needs no GPU either
    env2 = {k: v for k, v in env.items() if k != "PLASSHIP_CLI_DRYRUN"}
    out = subprocess.run([w, "concatdbs", "a", "b", "c", "--take-larger-entry"], env=env2, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert out.returncode == 0 and "args=concatdbs a b c --take-larger-entry" in out.stdout
    assert log.read_text().splitlines()[3].startswith("reference  <- plass-hip exit 95")


def test_deep_chain_fixtures_are_complete():
    """tests/golden/deep_chains.json (CPU oracle only, tests/golden/make_deep_chains.py): configs[1] as stated with its six iterations and the
    findassemblystart pass, twelve iterations of the community chain, six nucleotide and four guided iterations — every DB with entries, bytes
    and digest; tests/golden/c2_chain_digests.json (bench.py --config c2's `verify`) is the oracle's chain without findassemblystart"""
    import json
    d = json.load(open(os.path.join(ROOT, "tests", "golden", "deep_chains.json")))
    assert "CPU oracle only" in d["made_by"]
    c2, c3, c5 = d["c2_exact"], d["c3_deep"], d["c5_deep"]
    assert (c2["pairs"], c2["iters"], c2["findassemblystart"], len(c2["iterations"])) == (500000, 6, True, 6)
    assert all(k in c2["iterations"][0] for k in ("pref_uncorrected", "aln_uncorrected", "corrected", "pref", "aln", "seq"))
    assert c3["config"] == "c3" and 2 * c3["pairs"] >= 2000000 and len(c3["iterations"]) == 12 and not c3["findassemblystart"]
    assert len(c5["nucl"]) == 6 and len(c5["guided"]) == 4 and 2 * c5["pairs"] >= 2000000
    for rows, keys in ((c2["iterations"], ("pref", "aln", "seq")), (c3["iterations"], ("pref", "aln", "seq")), (c5["nucl"], ("pref", "aln", "assembly", "cycle", "rest")),
                       (c5["guided"], ("pref", "aln", "aln_nucl", "nucl", "aa"))):
        for r in rows:
            for k in keys:
                assert (r[k]["entries"] > 0 or k == "cycle") and r[k]["bytes"] >= 0 and re.fullmatch(r"[0-9a-f]{16}", r[k]["digest"]), (k, r[k])   # (no circular contig in this community)
    # residues grow along the chains (the chains were really chained)
    seqb = [r["seq"]["bytes"] for r in c3["iterations"]]
    assert seqb == sorted(seqb) and seqb[-1] > 2 * seqb[0]
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "c2_chain_digests.json")))
    assert "CPU oracle" in g["made_by"] and g["pairs"] == 500000 and g["digests"] == [r["seq"]["digest"] for r in d["c2_bench"]["iterations"]]
    assert g["digests"][0] != c2["iterations"][0]["seq"]["digest"]      # findassemblystart changes iteration 0 already


def test_deep_chain_fixtures_carry_the_reference_pin():
    """profiles/r05_deep_pin_reference.txt is the output of tests/golden/pin_deep_chains_against_reference.py: the unmodified reference's DBs,
    module by module, against tests/golden/deep_chains.json.  The record must cover EVERY DB of the committed fixture with the committed digest
    (a regenerated fixture without a new pin run fails here), and report no difference."""
    import json
    d = json.load(open(os.path.join(ROOT, "tests", "golden", "deep_chains.json")))
    rec = open(os.path.join(ROOT, "profiles", "r05_deep_pin_reference.txt")).read()
    sect = {}
    G = lambda f: json.load(open(os.path.join(ROOT, "tests", "golden", f)))
    d = dict(d, large_chain=G("large_chain.json"), large_nucl=G("large_nucl.json"))      # the older oracle-made fixtures, pinned by the same script
    for name in ("c2_exact", "c3_deep", "c5_deep", "c2_bench", "large_chain", "large_nucl"):
        body = rec.split("---- %s ----" % name)[1].split("\n(%s done" % name)[0]
        sect[name] = re.findall(r"^MATCH +(.*?) +entries +(\d+) bytes +(\d+) digest ([0-9a-f]{16})$", body, re.M)
        assert not re.search(r"^DIFFERS", body, re.M), name
        if not name.startswith("large"):                    # (the large sections are cut from a run that went on to another fixture: their lines are all there is)
            assert re.search(r"^DBs compared: (\d+), identical to the reference's: \1, differing: 0$", rec.split("(%s done" % name)[1], re.M), name

    def want(name, what, e):
        hit = [m for m in sect[name] if m[0] == what]
        assert len(hit) == 1 and (int(hit[0][1]), int(hit[0][2]), hit[0][3]) == (e["entries"], e["bytes"], e["digest"]), (name, what, hit, e)

    for name in ("c2_exact", "c3_deep", "c2_bench", "large_chain"):
        f = d[name]
        want(name, "synthetic reads", f["reads"]); want(name, "extractorfs x2 + translatenucs x2 + concatdbs", f["fragments"])
        for it, r in enumerate(f["iterations"]):
            want(name, "it %d: kmermatcher" % it, r["pref"]); want(name, "it %d: rescorediagonal" % it, r["aln"]); want(name, "it %d: assembleresults" % it, r["seq"])
        n = 2 + 3 * len(f["iterations"])
        if f.get("findassemblystart"):
            r = f["iterations"][0]
            want(name, "it 0: kmermatcher before findassemblystart", r["pref_uncorrected"]); want(name, "it 0: rescorediagonal before findassemblystart", r["aln_uncorrected"])
            want(name, "it 0: findassemblystart", r["corrected"])
            n += 3
        assert len(sect[name]) == n
    for name in ("c5_deep", "large_nucl"):
        f = d[name]
        want(name, "synthetic reads", f["reads"])
        want(name, "guided input: extractorfs x2 + concatdbs", f["guided_input"]["nucl"]); want(name, "guided input: translatenucs --add-orf-stop", f["guided_input"]["aa"])
        for it, r in enumerate(f["nucl"]):
            for k in ("pref", "aln", "assembly", "cycle", "rest"):
                want(name, "nucleotide it %d: %s" % (it, k), r[k])
        for it, r in enumerate(f["guided"]):
            for k in ("pref", "aln", "aln_nucl", "nucl", "aa"):
                want(name, "guided it %d: %s" % (it, k), r[k])
        assert len(sect[name]) == 3 + 5 * len(f["nucl"]) + 5 * len(f["guided"])
    # sequence data beyond 2^32 bytes (big_offsets.json)
    f = G("big_offsets.json")
    body = rec.split("---- big_offsets ----")[1].split("\n(big_offsets done")[0]
    sect["big_offsets"] = re.findall(r"^MATCH +(.*?) +entries +(\d+) bytes +(\d+) digest ([0-9a-f]{16})$", body, re.M)
    assert not re.search(r"^DIFFERS", body, re.M)
    want("big_offsets", "extractorfs x2 + translatenucs x2 + concatdbs (the live fragments)", f["live"]); want("big_offsets", "filler DB (numpy generator)", f["filler_db"])
    want("big_offsets", "concatdbs filler live: the DB beyond 2^32 bytes", f["db"])
    for it, r in enumerate(f["iterations"]):
        want("big_offsets", "it %d: kmermatcher" % it, r["pref"]); want("big_offsets", "it %d: rescorediagonal" % it, r["aln"]); want("big_offsets", "it %d: assembleresults" % it, r["seq"])
    assert f["db"]["bytes"] > 1 << 32 and len(sect["big_offsets"]) == 12
    assert sum(len(v) for v in sect.values()) == 134 + 11 + 28 + 12


def test_bench_workload_digests_carry_the_reference_pin():
    """tests/golden/c5_chain_digests.json (what `bench.py --config c5` verifies every run against; made by the GPU path) is identical to what the
    unmodified reference computes for the same 20 M reads: profiles/r05_headline_pin_reference.txt must show every one of the ten committed digests
    as a MATCH of the reference's run, and no difference"""
    import json
    h = json.load(open(os.path.join(ROOT, "tests", "golden", "c5_chain_digests.json")))
    rec = open(os.path.join(ROOT, "profiles", "r05_headline_pin_reference.txt")).read()
    body = rec.split("---- c5_headline ----")[1].split("---- c3_headline ----")[0]
    got = dict((m[0], m[1]) for m in re.findall(r"^MATCH +(.*?) +entries +\d+ bytes +\d+ digest ([0-9a-f]{16})$", body, re.M))
    assert not re.search(r"^DIFFERS", body, re.M) and "NOTE: the reference's kmermatcher split" not in body
    gd = [d.split("+") for d in h["digests"] if "+" in d]
    nu = [d for d in h["digests"] if "+" not in d]
    assert len(gd) == 5 and len(nu) == 5 and len(got) == 15
    for it, (a, b) in enumerate(gd):
        assert got["guided it %d: nucl" % it] == a and got["guided it %d: aa" % it] == b
    for it, d in enumerate(nu):
        assert got["nucleotide it %d: rest" % it] == d
    assert re.search(r"^DBs compared: 15, identical to the reference's: 15, differing: 0$", body.split("---- c3_headline ----")[0], re.M)
    # THE BENCH LINE'S WORKLOAD (50 M reads): tests/golden/c3_chain_digests.json, GPU-made; the reference follows the whole chain UNSPLIT
    # (--split-memory-limit 64G on the 62 GB build container) — all twelve digests it produced must be the committed ones
    h = json.load(open(os.path.join(ROOT, "tests", "golden", "c3_chain_digests.json")))
    body = rec.split("---- c3_headline ----")[1].split("# ---- the first run")[0]
    got = re.findall(r"^MATCH +it (\d+): assembleresults +entries +(\d+) bytes +\d+ digest ([0-9a-f]{16})$", body, re.M)
    assert not re.search(r"^DIFFERS", body, re.M) and "NOTE: the reference's kmermatcher split" not in body
    assert [int(it) for it, _, _ in got] == list(range(12)) and len(h["digests"]) == 12
    for it, n, dg in got:
        assert int(n) == h["fragments"] and dg == h["digests"][int(it)]
    assert re.search(r"^MATCH +extractorfs x2 \+ translatenucs x2 \+ concatdbs +entries +%d " % h["fragments"], body, re.M)
    assert re.search(r"^DBs compared: 13, identical to the reference's: 13, differing: 0$", body, re.M)
    # the first run (default memory limit) stopped where the reference began to split, and claims nothing beyond
    first = rec.split("# ---- the first run")[1]
    assert first.index("it 5: assembleresults") < first.index("NOTE: the reference's kmermatcher split") and "it 6: assembleresults" not in first


def test_scaling_model_and_furthest_below_of_the_bench_line():
    """bench.py: the cost model the N > 1 line prints (owner-filtered extraction up to 4 ranks, the exchange of level-1
    lines beyond) and the `roofline.furthest_below` selection"""
    import bench
    m1, m2, m4, m8 = (bench.scaling_model(w, 50e6) for w in (1, 2, 4, 8))
    assert m1["library_default"] == m2["library_default"] == m4["library_default"] == "owner_filtered" and m8["library_default"] == "exchange"
    assert m2["owner_filtered"]["total_ms"] < m2["exchange"]["total_ms"]            # one link per pair: replicated extraction beats moving every record
    assert m8["exchange"]["total_ms"] < m8["owner_filtered"]["total_ms"]            # seven links per GPU: the all-to-all wins
    assert 1.0 < m2["owner_filtered"]["speedup"] < m4["owner_filtered"]["speedup"] < m8["exchange"]["speedup"] < 8.0
    assert bench.scaling_model(4, 25e6)["single_gpu_ms"] == pytest.approx(m4["single_gpu_ms"] / 2, rel=1e-3)      # linear in the reads
    # furthest_below: the single kernel with the lowest fraction of the HBM peak by algorithmic bytes among those above 2 % of the step
    tot = {"kmermatcher_stage": [200.0, 280e9, False, 1], "rescore_stage": [30.0, 45e9, False, 1], "assemble_stage": [70.0, 51e9, False, 1],
           "fast": [50.0, 200e9, True, 1], "slow": [20.0, 5e9, True, 1], "tiny": [1.0, 1e6, True, 1], "stage_not_single": [90.0, 1e9, False, 1]}
    fb = bench.furthest_below(tot, 1, "no-such-config")
    assert fb["kernel"] == "slow" and fb["frac"] == pytest.approx(5e9 / 20e-3 / 1e9 / bench.HBM_PEAK_GBS) and fb["traffic_ratio"] is None
