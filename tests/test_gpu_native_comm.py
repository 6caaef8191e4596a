"""The library's own RCCL communicator (include/plasship_rccl.h) between REAL ranks on one GPU (run with `-m gpu`): W > 1 ranks run as a
child process whose `librccl` is tests/tools/rccl_stub.cpp (W threads, W contexts on this GPU) — comm_rccl.hip's exchange() with its
piece rounds, offsets and ncclSend / ncclRecv groups, the pinned ncclAllGather of the host arrays and the status rounds execute between
the ranks; only the wire is replaced.  Protein, nucleotide, guided and 130 k-fragment synthetic chains with 2, 3 and 4 ranks; every rank's
DB equals the reference's.

A module of its own (round 5; until then the last tests of tests/test_gpu_sharded.py): the child process needs device memory of its own —
its contexts' arenas and the kernels' scratch, which the HIP runtime allocates on demand — and that module's four long-lived contexts
have taken most of the HBM by the time these tests run (the arena takes one slab of most of the free HBM once a process has allocated
8 GB: the 1 M-read test does).  Twice in round 5's evidence runs the child died in a kernel launch with "out of resources, available
free memory 96 MB".  Here the parent holds one small context."""
import os

import pytest

from conftest import assert_same_db
from test_gpu_parity import km_params

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["exchange", "owner-filtered"])
def shard_extract_mode(request, monkeypatch):
    """both ways the k-mer records can reach the owner of their level-1 bucket (kmermatch.hip, shardOwnerFiltered; see tests/test_gpu_sharded.py)"""
    monkeypatch.setenv("PLASSHIP_TUNE_SHARD_EXTRACT", "1" if request.param == "exchange" else "2")
    return request.param


@pytest.fixture(scope="module")
def ref_ctx():
    import plass_amd
    c = plass_amd.Context(0)
    yield c
    c.close()


def test_sharded_one_rank_torch_rccl(tmp_path, golden):
    """the torch.distributed communicator bench.py uses (RCCL on device pointers of the library), in a 1-rank group:
    the same all-to-all(v) / all-gather calls an 8-GPU run makes, with this rank as its own peer"""
    import subprocess, sys
    code = r'''
import os, sys
sys.path.insert(0, %r)
import torch, torch.distributed as dist
import plass_amd
from plass_amd.shard import TorchComm
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
ctx = plass_amd.Context(0)
s = %r
def chain(ctx, tag):
    db = ctx.read_seqdb(s + "/seq_0")
    for it in range(2):
        par = plass_amd.KmermatchParams(hash_shift=67 if it == 0 else 68, include_only_extendable=(it > 0))
        c, _ = ctx.kmermatcher(db, par)
        a, _ = ctx.rescorediagonal(db, db, c, plass_amd.RescoreParams(min_seq_id=0.9))
        db, _ = ctx.assembleresults(db, a, plass_amd.AssembleParams(min_seq_id=0.9))
        db.write(%r + "/" + tag + "_seq_%%d" %% (it + 1))
comm = TorchComm(dist, torch.device("cuda:0"))
comm.install(ctx)
chain(ctx, "rccl")
assert comm.bytes_moved > 0
dist.destroy_process_group()
print("RCCL_OK", comm.bytes_moved)
''' % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.join(golden, "aa"), str(tmp_path))
    p = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert p.returncode == 0 and "RCCL_OK" in p.stdout, p.stdout[-3000:]
    for it in range(2):
        assert_same_db(os.path.join(golden, "aa", f"seq_{it + 1}"), tmp_path / f"rccl_seq_{it + 1}", f"1-rank RCCL iteration {it}")


def test_sharded_one_rank_native_rccl(tmp_path, golden):
    """the native communicator (include/plasship_rccl.h: RCCL loaded by the library, ncclSend / ncclRecv groups on the context's
    stream) in a 1-rank communicator: the sharded code path of all three modules, output = the reference's golden DBs"""
    import subprocess, sys
    code = r'''
import os, sys
sys.path.insert(0, %r)
import plass_amd
from plass_amd.shard import RcclComm, rccl_unique_id
ctx = plass_amd.Context(0)
s = %r
comm = RcclComm(ctx, 0, 1, rccl_unique_id())
db = ctx.read_seqdb(s + "/seq_0")
for it in range(2):
    par = plass_amd.KmermatchParams(hash_shift=67 if it == 0 else 68, include_only_extendable=(it > 0))
    c, _ = ctx.kmermatcher(db, par)
    a, _ = ctx.rescorediagonal(db, db, c, plass_amd.RescoreParams(min_seq_id=0.9))
    db, _ = ctx.assembleresults(db, a, plass_amd.AssembleParams(min_seq_id=0.9))
    db.write(%r + "/native_seq_%%d" %% (it + 1))
b, sec, n = comm.stats()
comm.destroy()
# back to single-GPU operation on the same context
db = ctx.read_seqdb(s + "/seq_0")
c, _ = ctx.kmermatcher(db, plass_amd.KmermatchParams(hash_shift=67, include_only_extendable=False))
print("NATIVE_RCCL_OK", b, n, c.count())
''' % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.join(golden, "aa"), str(tmp_path))
    p = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert p.returncode == 0 and "NATIVE_RCCL_OK" in p.stdout, p.stdout[-3000:]
    assert int(p.stdout.split("NATIVE_RCCL_OK")[1].split()[1]) > 0            # collectives were called
    for it in (1, 2):
        assert_same_db(os.path.join(golden, "aa", "seq_%d" % it), tmp_path / ("native_seq_%d" % it), "native RCCL communicator, iteration %d" % (it - 1))


STUB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "librccl_stub.so")


def _native_ranks(world, golden, out, case):
    """W > 1 ranks of the NATIVE communicator in a child process whose librccl is tests/tools/rccl_stub.cpp (W threads, W contexts
    on this GPU): comm_rccl.hip's exchange() — piece rounds, offsets, ncclSend / ncclRecv groups, the ncclAllGather of the host
    arrays and the status rounds — runs between real ranks; only the wire is replaced"""
    import subprocess, sys
    assert os.path.exists(STUB), "tests/tools/librccl_stub.so is missing: run `python __graft_entry__.py` (build()) first"
    env = dict(os.environ, PLASSHIP_RCCL_LIB=STUB, RCCL_STUB_STATS="1")
    p = subprocess.run([sys.executable, os.path.join(os.path.dirname(STUB), "native_comm_ranks.py"), str(world), str(golden), str(out), case],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900, env=env)
    assert p.returncode == 0 and "NATIVE_OK" in p.stdout, p.stdout[-4000:]
    sent, calls = (int(x) for x in p.stdout.split("NATIVE_OK")[1].split()[:2])
    assert calls > 0 and sent > 0, "no bytes crossed between the ranks: " + p.stdout[-2000:]
    assert "rccl_stub: group of %d ranks" % world in p.stdout                # the stub — not a real librccl — carried the run
    return p.stdout


@pytest.mark.timeout(1200)
@pytest.mark.parametrize("world", [2, 3, 4])
def test_native_comm_multi_rank_golden_aa(golden, tmp_path, world):
    """three protein iterations on the bundled example through plasship_rccl_comm with 2 / 3 / 4 ranks: every rank ends every
    iteration with the reference's DB; the ranks' candidate and alignment DBs are disjoint by query and their union is the
    reference's"""
    from conftest import read_db
    _native_ranks(world, golden, tmp_path, "aa")
    for it in range(3):
        for r in range(world):
            assert_same_db(os.path.join(golden, "aa", "seq_%d" % (it + 1)), tmp_path / ("r%d_seq_%d" % (r, it + 1)), "native comm, %d ranks, rank %d, iteration %d" % (world, r, it))
        for name in ("pref", "aln"):
            t, want = read_db(os.path.join(golden, "aa", "%s_%d" % (name, it)))
            union = {}
            for r in range(world):
                _, part = read_db(tmp_path / ("r%d_%s_%d" % (r, name, it)))
                for k, v in part.items():
                    if v not in (b"\0", b""):
                        assert k not in union or union[k] in (b"\0", b""), "query %d has lines on two ranks" % k
                        union[k] = v
                    else:
                        union.setdefault(k, v)
            bad = [k for k in want if union.get(k) != want[k]]
            assert not bad, "%s_%d: %d entries differ from the reference (first key %d: %r vs %r)" % (name, it, len(bad), bad[0], union.get(bad[0]), want[bad[0]])


@pytest.mark.timeout(1200)
@pytest.mark.parametrize("world", [2, 3, 4])
def test_native_comm_multi_rank_synthetic(ref_ctx, tmp_path, golden, world):
    """130 k protein fragments (two partition levels; 2 and 4 ranks exchange level-1 buckets, 3 ranks take the separate owner
    pass), three iterations through the native communicator against the single-context run"""
    import plass_amd
    from plass_amd import synth
    data, off, elen, key = synth.protein_fragment_db(40000, seed=5)
    ref = ref_ctx
    rdb = ref.upload_seqdb(data, off, elen, key, 0)
    for it in range(3):
        c, _ = ref.kmermatcher(rdb, km_params(it))
        a, _ = ref.rescorediagonal(rdb, rdb, c, plass_amd.RescoreParams(min_seq_id=0.9))
        rdb, _ = ref.assembleresults(rdb, a, plass_amd.AssembleParams(min_seq_id=0.9))
        rdb.write(tmp_path / f"e_seq_{it + 1}")
    _native_ranks(world, golden, tmp_path, "synthetic")
    for it in range(3):
        for r in range(world):
            assert_same_db(tmp_path / f"e_seq_{it + 1}", tmp_path / f"r{r}_seq_{it + 1}", f"native comm synthetic, rank {r} of {world}, iteration {it}")


@pytest.mark.timeout(1200)
def test_native_comm_multi_rank_nucl_and_guided(golden, tmp_path):
    """nucleotide chain (2 ranks) and the protein-guided chain (3 ranks) through the native communicator"""
    _native_ranks(2, golden, tmp_path, "nucl")
    for it in range(2):
        for r in range(2):
            assert_same_db(os.path.join(golden, "nucl", f"seq_{it + 1}"), tmp_path / f"r{r}_seq_{it + 1}", f"native comm nucl rank {r} it{it}")
    _native_ranks(3, golden, tmp_path, "guided")
    for it in range(2):
        for r in range(3):
            assert_same_db(os.path.join(golden, "guided", f"nucl_{it + 1}"), tmp_path / f"r{r}_nucl_{it + 1}", f"native comm guided nucl rank {r} it{it}")
            assert_same_db(os.path.join(golden, "guided", f"aa_{it + 1}"), tmp_path / f"r{r}_aa_{it + 1}", f"native comm guided aa rank {r} it{it}")
