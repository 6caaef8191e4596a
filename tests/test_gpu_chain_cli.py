"""GPU tests of the fused C++ drivers of plass-hip (`-m gpu`): `assemble-chain`, `nuclassemble-chain`, `guidedassemble-chain` run the
iteration loops of data/assemble.sh:85-156, data/nuclassemble.sh:95-137 and data/guidedNuclAssemble.sh:77-126 with every DB of the
loop resident in HBM — one DB read, one written.  The bar: the final DB equals the reference's golden DB where the reference wrote one
for exactly this chain, else the DB the per-module calls (each pinned on the reference by tests/test_gpu_parity.py) produce."""
import os
import subprocess

import pytest

from conftest import assert_same_db
from test_gpu_parity import gd_km_params, gd_rs_params, km_params, nucl_as_params

pytestmark = pytest.mark.gpu
HIP = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "plass_amd", "plass-hip")


def run(args):
    p = subprocess.run([HIP] + [str(a) for a in args], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert p.returncode == 0 and "Time for processing" in p.stdout, p.stdout[-3000:]
    return p.stdout


@pytest.fixture(scope="module")
def ctx():
    import plass_amd
    c = plass_amd.Context(0)
    yield c
    c.close()


def test_assemble_chain_iteration0_is_the_reference_workflow(golden, tmp_path):
    """one iteration = kmermatcher, rescorediagonal, findassemblystart, kmermatcher, rescorediagonal, assembleresults (data/assemble.sh:
    85-156): the reference's assembly_0 of the bundled example — from the fragment DB, and from the read DB through the GPU preprocessing"""
    run(["assemble-chain", os.path.join(golden, "aa", "seq_0"), tmp_path / "a0", "--num-iterations", "1"])
    assert_same_db(os.path.join(golden, "fs", "assembly_0"), tmp_path / "a0", "assemble-chain, 1 iteration, from aa_6f_start_long")
    run(["assemble-chain", os.path.join(golden, "nucl", "seq_0"), tmp_path / "a0r", "--num-iterations", "1", "--from-reads", "1"])
    assert_same_db(os.path.join(golden, "fs", "assembly_0"), tmp_path / "a0r", "assemble-chain, 1 iteration, from the reads")


def test_assemble_chain_equals_the_module_calls(ctx, golden, tmp_path):
    """four iterations with --write-intermediate (a writer thread on a second context while the next iteration runs): every
    intermediate DB and the final DB equal what the per-module calls chain to"""
    import plass_amd
    rs, asp = plass_amd.RescoreParams(min_seq_id=0.9), plass_amd.AssembleParams(min_seq_id=0.9)
    db = ctx.read_seqdb(os.path.join(golden, "aa", "seq_0"))
    hs = 67
    for it in range(4):
        hs += it % 2
        kp = plass_amd.KmermatchParams(hash_shift=hs, include_only_extendable=(it > 0))
        c, _ = ctx.kmermatcher(db, kp); a, _ = ctx.rescorediagonal(db, db, c, rs)
        if it == 0:
            db, _ = ctx.findassemblystart(db, a)
            c, _ = ctx.kmermatcher(db, kp); a, _ = ctx.rescorediagonal(db, db, c, rs)
        db, _ = ctx.assembleresults(db, a, asp)
        db.write(tmp_path / f"e_{it}")
    inter = tmp_path / "inter"; inter.mkdir()
    out = run(["assemble-chain", os.path.join(golden, "aa", "seq_0"), tmp_path / "final", "--num-iterations", "4", "--write-intermediate", inter])
    assert "chain: 4 iterations" in out
    for it in range(3):
        assert os.path.exists(inter / f"assembly_{it}.done")
        assert_same_db(tmp_path / f"e_{it}", inter / f"assembly_{it}", f"assemble-chain, intermediate DB of iteration {it}")
    assert_same_db(tmp_path / "e_3", tmp_path / "final", "assemble-chain, final DB")


def test_nuclassemble_and_guided_chains(ctx, golden, tmp_path):
    import plass_amd
    s = os.path.join(golden, "nucl")
    db = ctx.read_seqdb(f"{s}/seq_0")
    for it in range(2):
        c, _ = ctx.kmermatcher(db, km_params(it, nucl=True)); a, _ = ctx.rescorediagonal(db, db, c, plass_amd.RescoreParams(min_seq_id=0.99))
        out, _ = ctx.assembleresults(db, a, nucl_as_params())
        cyc, db, _ = ctx.cyclecheck(out, max_seq_len=200000, chop_cycle=True, with_rest=True)
    db.write(tmp_path / "e_nucl")
    run(["nuclassemble-chain", f"{s}/seq_0", tmp_path / "nucl", "--num-iterations", "2"])
    assert_same_db(tmp_path / "e_nucl", tmp_path / "nucl", "nuclassemble-chain")
    # guided: reads -> ORFs + twins (data/guidedNuclAssemble.sh:44-75) -> 2 iterations, against the per-module calls on the same inputs
    reads = ctx.read_seqdb(f"{s}/seq_0")
    nu, aa = ctx.penguin_guided_inputs(reads)
    for it in range(2):
        c, _ = ctx.kmermatcher(aa, gd_km_params()); a, _ = ctx.rescorediagonal(aa, aa, c, gd_rs_params())
        na, _ = ctx.proteinaln2nucl(nu, aa, a)
        nu, aa, _ = ctx.guidedassembleresults(nu, aa, na)
    nu.write(tmp_path / "e_gn"); aa.write(tmp_path / "e_ga")
    run(["guidedassemble-chain", f"{s}/seq_0", tmp_path / "gn", tmp_path / "ga", "--num-iterations", "2"])
    assert_same_db(tmp_path / "e_gn", tmp_path / "gn", "guidedassemble-chain, nucleotide ORFs")
    assert_same_db(tmp_path / "e_ga", tmp_path / "ga", "guidedassemble-chain, protein twins")
