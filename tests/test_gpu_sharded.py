"""GPU tests of ONE read set sharded over several ranks (include/plasship.h: plasship_ctx_set_comm).

W ranks run as W threads with W contexts on the one GPU of the box (plass_amd.shard.LocalGroup: the collectives are
device-to-device copies), i.e. exactly the code path of a multi-GPU run: extraction of a share of the sequences,
partition by owner + all-to-all(v) of the k-mer records and of the grouped records, the halo of the reference's run scan
across rank boundaries, owned-query rescoring / extension and the all-gather of the extended sequences.  The bar is the
single-GPU bar: every rank's output DB equals the reference's golden DB (or the single-context result, which the other
tests pin on the reference), and the union of the ranks' candidate / alignment lists equals the single-context list.
"""
import os

import numpy as np
import pytest

from conftest import assert_same_db
from test_gpu_parity import gd_km_params, gd_rs_params, km_params, nucl_as_params

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["exchange", "owner-filtered"])
def shard_extract_mode(request, monkeypatch):
    """every test of this file runs both ways the k-mer records can reach the owner of their level-1 bucket (kmermatch.hip,
    shardOwnerFiltered): the all-to-all of level-1 lines (rounds 1-4), and the reference's own MPI scheme — every rank extracts all
    sequences and keeps its hash range (kmermatcher.cpp:312,736-778), no exchange 1 (round 5; the default up to 4 ranks)"""
    monkeypatch.setenv("PLASSHIP_TUNE_SHARD_EXTRACT", "1" if request.param == "exchange" else "2")
    return request.param


@pytest.fixture(scope="module")
def ctxs():
    import plass_amd
    cs = [plass_amd.Context(0) for _ in range(4)]
    yield cs
    for c in cs:
        c.close()


def _run(ctxs, world, fn):
    from plass_amd.shard import LocalGroup
    return LocalGroup(world).run(fn, ctxs[:world])


def _cands_rows(c):
    q, t, s, d = c.download()
    return np.stack([q.astype(np.int64), t.astype(np.int64), s.astype(np.int64), d.astype(np.int64)], axis=1)


def _aln_rows(a):
    return [(r.query_key, r.target_key, r.bit_score, r.raw_score, r.seq_id, r.q_start, r.q_end, r.q_len, r.db_start, r.db_end, r.db_len,
             r.aln_len, r.reversed) for r in a.download()]


def _check_union(per_rank, single, what):
    """ranks own ascending id ranges and a query's lines live on one rank: concatenation in rank order == single list"""
    if isinstance(single, np.ndarray):
        u = np.concatenate(per_rank, axis=0)
        assert u.shape == single.shape and (u == single).all(), what
    else:
        u = [x for part in per_rank for x in part]
        assert u == single, what


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_golden_aa_chained(ctxs, golden, tmp_path, world):
    """three protein iterations on the bundled example: every rank ends every iteration with the reference's DB"""
    import plass_amd
    s = os.path.join(golden, "aa")
    ref = ctxs[3]
    rdb = ref.read_seqdb(f"{s}/seq_0")
    expect = []
    for it in range(3):
        c, _ = ref.kmermatcher(rdb, km_params(it))
        a, _ = ref.rescorediagonal(rdb, rdb, c, plass_amd.RescoreParams(min_seq_id=0.9))
        expect.append((_cands_rows(c), _aln_rows(a)))
        rdb, _ = ref.assembleresults(rdb, a, plass_amd.AssembleParams(min_seq_id=0.9))

    def work(rank, ctx):
        db = ctx.read_seqdb(f"{s}/seq_0")
        out = []
        for it in range(3):
            cands, kst = ctx.kmermatcher(db, km_params(it))
            alns, _ = ctx.rescorediagonal(db, db, cands, plass_amd.RescoreParams(min_seq_id=0.9))
            out.append((_cands_rows(cands), _aln_rows(alns), kst.n_candidates))
            db2, _ = ctx.assembleresults(db, alns, plass_amd.AssembleParams(min_seq_id=0.9))
            db2.write(tmp_path / f"r{rank}_seq_{it + 1}")
            db = db2
        return out

    res = _run(ctxs, world, work)
    for it in range(3):
        _check_union([res[r][it][0] for r in range(world)], expect[it][0], f"candidates it{it}")
        _check_union([res[r][it][1] for r in range(world)], expect[it][1], f"alignments it{it}")
        assert sum(res[r][it][2] for r in range(world)) == int((expect[it][0][:, 0] != expect[it][0][:, 1]).sum())
        for r in range(world):
            assert_same_db(f"{s}/seq_{it + 1}", tmp_path / f"r{r}_seq_{it + 1}", f"rank {r} of {world}, iteration {it}")


@pytest.mark.parametrize("case", [1, 2, 3, 4])
def test_sharded_stale_scan_quirk(ctxs, golden, tmp_path, case):
    """the reference's scan into stale records behind the compaction point, when the records are spread over ranks:
    the ranks reduce N_m / N_k / the sort-#1 ranks, the owner of the last run appends the stale records"""
    d = os.path.join(golden, "q1", f"case{case}")
    ext = open(os.path.join(d, "ext")).read().strip() == "1"
    par = km_params(0); par.include_only_extendable = ext
    ref = ctxs[3]
    rdb = ref.read_seqdb(os.path.join(d, "seq"))
    c, _ = ref.kmermatcher(rdb, par)
    c.write(tmp_path / "pref")
    assert_same_db(os.path.join(d, "pref"), tmp_path / "pref", "single context")
    expect = _cands_rows(c)
    for world in (2, 3):
        res = _run(ctxs, world, lambda rank, ctx: _cands_rows(ctx.kmermatcher(ctx.read_seqdb(os.path.join(d, "seq")), par)[0]))
        _check_union(res, expect, f"stale-scan case {case}, {world} ranks")


def test_sharded_golden_nucl_chained(ctxs, golden, tmp_path):
    """nucleotide path (canonical k-mers, strand bits, first-run quirk key reduced over the ranks, nuclassembleresults)"""
    import plass_amd
    s = os.path.join(golden, "nucl")
    world = 2

    def work(rank, ctx):
        db = ctx.read_seqdb(f"{s}/seq_0")
        for it in range(2):
            cands, _ = ctx.kmermatcher(db, km_params(it, nucl=True))
            alns, _ = ctx.rescorediagonal(db, db, cands, plass_amd.RescoreParams(min_seq_id=0.99))
            db2, _ = ctx.assembleresults(db, alns, nucl_as_params())
            db2.write(tmp_path / f"r{rank}_seq_{it + 1}")
            db = db2

    _run(ctxs, world, work)
    for it in range(2):
        for r in range(world):
            assert_same_db(f"{s}/seq_{it + 1}", tmp_path / f"r{r}_seq_{it + 1}", f"nucl rank {r}, iteration {it}")


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_nucleotide_strand_ties(ctxs, golden, tmp_path, world):
    """the strand of a mixed-strand triple when its records were grouped on DIFFERENT ranks: the triples travel with the rank word of
    their top member (TripleX, kmermatch.hip) and the owner keeps the larger one — the reference's pref DB from every rank's share"""
    s = os.path.join(golden, "strand_ties")
    ref = ctxs[3]
    expect = _cands_rows(ref.kmermatcher(ref.read_seqdb(f"{s}/seq_0"), km_params(0, nucl=True))[0])
    ref.kmermatcher(ref.read_seqdb(f"{s}/seq_0"), km_params(0, nucl=True))[0].write(tmp_path / "single")
    assert_same_db(f"{s}/pref_0", tmp_path / "single", "strand ties, single context")
    res = _run(ctxs, world, lambda rank, ctx: _cands_rows(ctx.kmermatcher(ctx.read_seqdb(f"{s}/seq_0"), km_params(0, nucl=True))[0]))
    _check_union(res, expect, f"strand ties, {world} ranks")


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_hairpin_genome_strand_ties(ctxs, tmp_path, world):
    """many mixed-strand triples (a genome with planted inverted repeats) whose records are grouped on different ranks: the union of
    the ranks' candidate lists equals the single-context list (which test_gpu_parity pins on the oracle)"""
    from plass_amd import synth
    reads, _ = synth.nucleotide_hairpin_reads(20000, 40, seed=23, coverage=30.0)
    data, off, elen, key = synth.fixed_length_db(reads)
    ref = ctxs[3]
    expect = _cands_rows(ref.kmermatcher(ref.upload_seqdb(data, off, elen, key, 1), km_params(0, nucl=True))[0])
    res = _run(ctxs, world, lambda rank, ctx: _cands_rows(ctx.kmermatcher(ctx.upload_seqdb(data, off, elen, key, 1), km_params(0, nucl=True))[0]))
    _check_union(res, expect, f"hairpin strand ties, {world} ranks")


def test_sharded_golden_long_nucl(ctxs, golden, tmp_path):
    """24-byte records (KmerPosition<int>), contigs of tens of kb, 3 ranks"""
    import plass_amd
    s = os.path.join(golden, "longnucl")
    world = 3

    def work(rank, ctx):
        db = ctx.read_seqdb(f"{s}/seq_0")
        cands, _ = ctx.kmermatcher(db, km_params(0, nucl=True))
        alns, _ = ctx.rescorediagonal(db, db, cands, plass_amd.RescoreParams(min_seq_id=0.99))
        db2, _ = ctx.assembleresults(db, alns, nucl_as_params())
        db2.write(tmp_path / f"r{rank}_seq_1")

    _run(ctxs, world, work)
    for r in range(world):
        assert_same_db(f"{s}/seq_1", tmp_path / f"r{r}_seq_1", f"long nucl rank {r}")


def test_sharded_golden_guided_chained(ctxs, golden, tmp_path):
    """protein-guided stage: two output DBs (nucleotide ORFs + protein twins) gathered from the ranks"""
    s = os.path.join(golden, "guided")
    world = 2

    def work(rank, ctx):
        aa = ctx.read_seqdb(f"{s}/aa_0"); nu = ctx.read_seqdb(f"{s}/nucl_0")
        for it in range(2):
            cands, _ = ctx.kmermatcher(aa, gd_km_params())
            alns, _ = ctx.rescorediagonal(aa, aa, cands, gd_rs_params())
            naln, _ = ctx.proteinaln2nucl(nu, aa, alns)
            nu2, aa2, _ = ctx.guidedassembleresults(nu, aa, naln)
            nu2.write(tmp_path / f"r{rank}_nucl_{it + 1}"); aa2.write(tmp_path / f"r{rank}_aa_{it + 1}")
            nu, aa = nu2, aa2

    _run(ctxs, world, work)
    for it in range(2):
        for r in range(world):
            assert_same_db(f"{s}/nucl_{it + 1}", tmp_path / f"r{r}_nucl_{it + 1}", f"guided nucl rank {r} it{it}")
            assert_same_db(f"{s}/aa_{it + 1}", tmp_path / f"r{r}_aa_{it + 1}", f"guided aa rank {r} it{it}")


def test_sharded_keep_target0(ctxs, golden, tmp_path):
    """--keep-target 0: the 'consumed as a target' bits are OR-ed over the ranks before the output DB is built"""
    import plass_amd
    s = os.path.join(golden, "aa")
    world = 3

    def work(rank, ctx):
        db = ctx.read_seqdb(f"{s}/seq_0")
        cands, _ = ctx.kmermatcher(db, km_params(0))
        alns, _ = ctx.rescorediagonal(db, db, cands, plass_amd.RescoreParams(min_seq_id=0.9))
        out, _ = ctx.assembleresults(db, alns, plass_amd.AssembleParams(min_seq_id=0.9, keep_target=False))
        out.write(tmp_path / f"r{rank}_seq")

    _run(ctxs, world, work)
    for r in range(world):
        assert_same_db(f"{s}/seq_1_keeptarget0", tmp_path / f"r{r}_seq", f"keep-target 0 rank {r}")


def test_sharded_adversarial_inputs(ctxs, golden, tmp_path):
    """hostile inputs (shorter than k, homopolymers, duplicates, one long contig, index order != key order) on 3 ranks:
    ranks with next to nothing to do, owners without records"""
    import plass_amd
    s = os.path.join(golden, "adversarial")
    for kind, nucl, rs, asp in (("aa", False, plass_amd.RescoreParams(min_seq_id=0.9), plass_amd.AssembleParams(min_seq_id=0.9)),
                                ("nucl", True, plass_amd.RescoreParams(min_seq_id=0.99), nucl_as_params())):
        src = f"{s}/{kind}_seq"
        ref = ctxs[3]
        rdb = ref.read_seqdb(src)
        c, _ = ref.kmermatcher(rdb, km_params(0, nucl=nucl))
        a, _ = ref.rescorediagonal(rdb, rdb, c, rs)
        o, _ = ref.assembleresults(rdb, a, asp)
        o.write(tmp_path / f"{kind}_expect")
        expect = _cands_rows(c)

        def work(rank, ctx):
            db = ctx.read_seqdb(src)
            cands, _ = ctx.kmermatcher(db, km_params(0, nucl=nucl))
            alns, _ = ctx.rescorediagonal(db, db, cands, rs)
            out, _ = ctx.assembleresults(db, alns, asp)
            out.write(tmp_path / f"{kind}_r{rank}")
            return _cands_rows(cands)

        res = _run(ctxs, 3, work)
        _check_union(res, expect, f"adversarial {kind} candidates")
        for r in range(3):
            assert_same_db(tmp_path / f"{kind}_expect", tmp_path / f"{kind}_r{r}", f"adversarial {kind} rank {r}")


def test_sharded_findassemblystart_iteration0(ctxs, golden, tmp_path):
    """iteration 0 of `plass assemble` (kmermatcher, rescorediagonal, findassemblystart, kmermatcher, rescorediagonal,
    assembleresults) on 3 ranks: the start votes of the owned queries are max-reduced over the ranks"""
    import plass_amd
    s, f = os.path.join(golden, "aa"), os.path.join(golden, "fs")
    rs, asp = plass_amd.RescoreParams(min_seq_id=0.9), plass_amd.AssembleParams(min_seq_id=0.9)

    def work(rank, ctx):
        db = ctx.read_seqdb(f"{s}/seq_0")
        cands, _ = ctx.kmermatcher(db, km_params(0))
        alns, _ = ctx.rescorediagonal(db, db, cands, rs)
        corr, _ = ctx.findassemblystart(db, alns)
        corr.write(tmp_path / f"r{rank}_corr")
        cands2, _ = ctx.kmermatcher(corr, km_params(0))
        alns2, _ = ctx.rescorediagonal(corr, corr, cands2, rs)
        as0, _ = ctx.assembleresults(corr, alns2, asp)
        as0.write(tmp_path / f"r{rank}_as0")

    _run(ctxs, 3, work)
    for r in range(3):
        assert_same_db(f"{f}/corrected_seqs", tmp_path / f"r{r}_corr", f"findassemblystart rank {r}")
        assert_same_db(f"{f}/assembly_0", tmp_path / f"r{r}_as0", f"iteration 0 rank {r}")


@pytest.mark.parametrize("world", [2, 3, 4])
def test_sharded_synthetic(ctxs, tmp_path, world):
    """40 k read pairs (130 k protein fragments, 7.8 M k-mer record slots: two partition levels), 3 iterations against the
    single-context run.  2 and 4 ranks take the folded path (level 1 doubles as the partition by owner), 3 ranks the separate
    owner pass"""
    import plass_amd
    from plass_amd import synth
    data, off, elen, key = synth.protein_fragment_db(40000, seed=5)
    ref = ctxs[3]
    rdb = ref.upload_seqdb(data, off, elen, key, 0)
    expect = []
    for it in range(3):
        c, _ = ref.kmermatcher(rdb, km_params(it))
        a, _ = ref.rescorediagonal(rdb, rdb, c, plass_amd.RescoreParams(min_seq_id=0.9))
        expect.append((_cands_rows(c), len(_aln_rows(a))))
        rdb, _ = ref.assembleresults(rdb, a, plass_amd.AssembleParams(min_seq_id=0.9))
        rdb.write(tmp_path / f"e_seq_{it + 1}")

    def work(rank, ctx):
        db = ctx.upload_seqdb(data, off, elen, key, 0)
        out = []
        for it in range(3):
            cands, _ = ctx.kmermatcher(db, km_params(it))
            alns, _ = ctx.rescorediagonal(db, db, cands, plass_amd.RescoreParams(min_seq_id=0.9))
            out.append((_cands_rows(cands), alns.count()))
            db2, _ = ctx.assembleresults(db, alns, plass_amd.AssembleParams(min_seq_id=0.9))
            db2.write(tmp_path / f"r{rank}_seq_{it + 1}")
            db = db2
        return out

    res = _run(ctxs, world, work)
    for it in range(3):
        _check_union([res[r][it][0] for r in range(world)], expect[it][0], f"synthetic candidates it{it}")
        assert sum(res[r][it][1] for r in range(world)) == expect[it][1]
        for r in range(world):
            assert_same_db(tmp_path / f"e_seq_{it + 1}", tmp_path / f"r{r}_seq_{it + 1}", f"synthetic rank {r} it{it}")


def test_sharded_eight_ranks_small_set(ctxs):
    """8 ranks on the 3 000-pair set bench.py uses as its multi-GPU preflight (about 1 200 fragments per rank: owners with a
    handful of records, empty exchanges): two iterations against the single-context run"""
    import plass_amd
    from plass_amd import synth
    data, off, elen, key = synth.protein_fragment_db(3000, seed=7)
    rs, asp = plass_amd.RescoreParams(min_seq_id=0.9), plass_amd.AssembleParams(min_seq_id=0.9)
    ref = ctxs[3]
    rdb = ref.upload_seqdb(data, off, elen, key, 0)
    expect = []
    for it in range(2):
        c, _ = ref.kmermatcher(rdb, km_params(it))
        a, _ = ref.rescorediagonal(rdb, rdb, c, rs)
        rdb, _ = ref.assembleresults(rdb, a, asp)
        expect.append((_cands_rows(c), rdb.download()[0]))
    eight = ctxs + [plass_amd.Context(0) for _ in range(4)]

    def work(rank, ctx):
        db = ctx.upload_seqdb(data, off, elen, key, 0)
        res = []
        for it in range(2):
            c, _ = ctx.kmermatcher(db, km_params(it))
            a, _ = ctx.rescorediagonal(db, db, c, rs)
            db, _ = ctx.assembleresults(db, a, asp)
            res.append((_cands_rows(c), db.download()[0]))
        return res

    try:
        got = _run(eight, 8, work)
    finally:
        for c in eight[4:]:
            c.close()
    for it in range(2):
        _check_union([got[r][it][0] for r in range(8)], expect[it][0], f"8 ranks, candidates it{it}")
        for r in range(8):
            assert got[r][it][1] == expect[it][1], f"8 ranks, rank {r}, contig DB it{it}"


def test_full_size_sharded_equals_single_and_properties(ctxs):
    """BASELINE.json configs[1] (1 M reads = 500 k pairs, 1.6 M protein fragments, ~1 GB of k-mer records): two iterations on
    2 ranks against the single-context run — candidate lists and contig DBs identical — and size-independent properties of
    the result: candidates sorted by (query, target) without self hits, accepted alignments a subset of the candidates,
    every sequence of the output contains its input sequence (greedy extension only appends), key set unchanged"""
    import plass_amd
    import bench
    # the bench workload of configs[1], made the way bench.py makes it: reads generated in HBM, fragments by the GPU preprocessing
    frag, _ = bench.build_workload(ctxs[3], "c2")
    data, off, elen, key = frag.download()
    frag.free()
    rs, asp = plass_amd.RescoreParams(min_seq_id=0.9), plass_amd.AssembleParams(min_seq_id=0.9)
    ref = ctxs[3]
    rdb = ref.upload_seqdb(data, off, elen, key, 0)
    expect = []
    for it in range(2):
        c, kst = ref.kmermatcher(rdb, km_params(it))
        a, rst = ref.rescorediagonal(rdb, rdb, c, rs)
        q, t, sc, dg = c.download()
        assert np.all((q[1:] > q[:-1]) | ((q[1:] == q[:-1]) & (t[1:] > t[:-1]))) and not np.any(q == t)
        assert kst.n_candidates == len(q) > 3_000_000 and len(key) <= rst.n_accepted <= len(q) + len(key)
        out, ast = ref.assembleresults(rdb, a, asp)
        d1, o1, e1, k1 = out.download()
        d0, o0, e0, k0 = rdb.download()
        assert np.array_equal(k0, k1) and np.all(e1 >= e0) and ast.n_extended > 100000
        # every output sequence contains the input sequence it grew from (checked on every 97th sequence)
        for i in range(0, len(k0), 97):
            s0 = d0[int(o0[i]):int(o0[i]) + int(e0[i]) - 2]; s1 = d1[int(o1[i]):int(o1[i]) + int(e1[i]) - 2]
            assert s0 in s1
        expect.append((q.tobytes(), t.tobytes(), sc.tobytes(), dg.tobytes(), d1))
        a.free(); c.free(); rdb.free(); rdb = out

    def work(rank, ctx):
        db = ctx.upload_seqdb(data, off, elen, key, 0)
        res = []
        for it in range(2):
            c, _ = ctx.kmermatcher(db, km_params(it))
            a, _ = ctx.rescorediagonal(db, db, c, rs)
            out, _ = ctx.assembleresults(db, a, asp)
            res.append((c.download(), out.download()[0]))
            a.free(); c.free(); db.free(); db = out
        db.free()
        return res

    got = _run(ctxs, 2, work)
    for it in range(2):
        for j in range(4):
            assert b"".join(got[r][it][0][j].tobytes() for r in range(2)) == expect[it][j], f"candidates differ, iteration {it}"
        for r in range(2):
            assert got[r][it][1] == expect[it][4], f"contig DB of rank {r} differs, iteration {it}"


@pytest.mark.timeout(300)
@pytest.mark.parametrize("nth", [0, 1, 3, 6, 8])
def test_sharded_rank_local_failure_ends_the_call_on_every_rank(ctxs, golden, tmp_path, nth, shard_extract_mode):
    """one rank fails on its own between two collectives (injected before its nth collective of the iteration): every rank must
    return an error from the SAME library call — the failing rank its own, the others PLASSHIP_ERR_PEER (-5) — instead of
    waiting inside the next collective for a rank that has left (this test would hang); the group stays usable: the next
    iteration, without injection, gives the reference's DB on every rank"""
    import plass_amd
    if shard_extract_mode == "owner-filtered" and nth >= 8:
        pytest.skip("an owner-filtered iteration has fewer collectives than this (no exchange 1 and its counts)")
    s = os.path.join(golden, "aa")
    world, bad = 3, 1

    def iteration(ctx, db):
        cands, _ = ctx.kmermatcher(db, km_params(0))
        alns, _ = ctx.rescorediagonal(db, db, cands, plass_amd.RescoreParams(min_seq_id=0.9))
        return ctx.assembleresults(db, alns, plass_amd.AssembleParams(min_seq_id=0.9))[0]

    def work(rank, ctx):
        db = ctx.read_seqdb(f"{s}/seq_0")
        if rank == bad:
            plass_amd._lib._check(ctx.lib.plasship_ctx_debug_fail_collective(ctx.h, nth), "plasship_ctx_debug_fail_collective")
        err = None
        try:
            iteration(ctx, db)
        except plass_amd.PlasshipError as e:                   # no barrier abort, no help from the harness: the library must get everybody out
            err = str(e)
        plass_amd._lib._check(ctx.lib.plasship_ctx_debug_fail_collective(ctx.h, -1), "plasship_ctx_debug_fail_collective")
        out = iteration(ctx, db)
        out.write(tmp_path / f"r{rank}_seq_1")
        return err

    errs = _run(ctxs, world, work)
    assert all(e is not None for e in errs), errs
    assert "injected rank-local failure" in errs[bad] and "(-3)" in errs[bad]
    for r in range(world):
        if r != bad:
            assert "(-5)" in errs[r] and "rank %d failed" % bad in errs[r], errs[r]
        assert_same_db(f"{s}/seq_1", tmp_path / f"r{r}_seq_1", f"rank {r} after the failed call")
