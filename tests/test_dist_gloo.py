"""world_size-2 gloo tests (CPU) of the multi-rank plumbing: the step reduction bench.py reports with (max of the times, sum of the
overlaps) and the communicator callbacks of the sharded run (plass_amd/shard.py: TorchComm) against the in-process reference."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    from plass_amd import dist as pd
    d = pd.init("gloo", rank, world)
    # step reduction: time = max, overlaps = sum
    t, c = pd.reduce_step(d, 0.010 * (rank + 1), 1000 + rank)
    q.put((rank, t, c))
    d.barrier()
    d.destroy_process_group()


def test_two_rank_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, t0, c0), (r1, t1, c1) = res
    assert t0 == t1 == pytest.approx(0.020) and c0 == c1 == 2001          # max time, summed overlaps on every rank


def _comm_worker(rank, world, port, q):
    """the communicator bench.py installs on a context (plass_amd.shard.TorchComm), driven through its C callbacks
    exactly as libplasship drives them, on host buffers over gloo"""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import ctypes as C
    import torch.distributed as dist
    from plass_amd.shard import TorchComm, owned_range
    dist.init_process_group("gloo", rank=rank, world_size=world)
    comm = TorchComm(dist, device=None)
    comm.CHUNK = 40           # several rounds of point-to-point pieces per message, a different number per pair
    cs = comm.struct
    # allgather_host: 3 u64 per rank
    send = (C.c_uint64 * 3)(rank, 10 + rank, 100 * (rank + 1)); recv = (C.c_uint64 * (3 * world))()
    assert cs.allgather_host(None, C.addressof(send), C.addressof(recv), 24) == 0
    gathered = list(recv)
    # alltoallv_dev: 16-byte records tagged (source, destination, i); rank r sends 5 + 3 r + d records to rank d
    cnt = [5 + 3 * rank + d for d in range(world)]
    rec = np.zeros((sum(cnt), 2), dtype=np.uint64)
    o = 0
    for d in range(world):
        for i in range(cnt[d]):
            rec[o] = (rank * 1000 + d, i); o += 1
    sb = (C.c_uint64 * world)(*[16 * c for c in cnt])
    rcnt = [5 + 3 * s + rank for s in range(world)]
    rb = (C.c_uint64 * world)(*[16 * c for c in rcnt])
    out = np.zeros((sum(rcnt), 2), dtype=np.uint64)
    assert cs.alltoallv_dev(None, rec.ctypes.data, sb, out.ctypes.data, rb) == 0
    ok = True; o = 0
    for s in range(world):
        for i in range(rcnt[s]):
            ok = ok and tuple(int(x) for x in out[o]) == (s * 1000 + rank, i); o += 1
    # allgatherv_dev: rank r contributes 7 + 5 r bytes of value r + 1
    mine = np.full(7 + 5 * rank, rank + 1, dtype=np.uint8)
    gb = (C.c_uint64 * world)(*[7 + 5 * r for r in range(world)])
    allb = np.zeros(sum(7 + 5 * r for r in range(world)), dtype=np.uint8)
    assert cs.allgatherv_dev(None, mine.ctypes.data, len(mine), allb.ctypes.data, gb) == 0
    expect = np.concatenate([np.full(7 + 5 * r, r + 1, dtype=np.uint8) for r in range(world)])
    q.put((rank, gathered, ok, bool((allb == expect).all()), owned_range(7, rank, world), comm.calls))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_sharded_communicator():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_comm_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, gathered, ok, okg, own, calls in res:
        assert gathered == [0, 10, 100, 1, 11, 200]
        assert ok and okg and calls == 3
    assert res[0][4] == (0, 4) and res[1][4] == (4, 7)          # ids [ceil(r n / W), ceil((r + 1) n / W))
