"""world_size-2 gloo test (CPU) of the multi-rank plumbing bench.py uses: rank partition plan, max/sum step
reduction, and the count + record all-to-all that the k-mer-bucket exchange is built on."""
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
    import torch
    from plass_amd import dist as pd
    d = pd.init("gloo", rank, world)
    plan = pd.partition_plan(world, n_buckets=64)
    # step reduction: time = max, overlaps = sum
    t, c = pd.reduce_step(d, 0.010 * (rank + 1), 1000 + rank)
    # bucketed exchange: every rank holds records tagged (src, bucket, i); after the all-to-all a rank holds exactly the
    # records of the buckets it owns, from every source
    rng = np.random.default_rng(100 + rank)
    counts = rng.integers(0, 50, size=64)
    recs = []
    for dst in range(world):
        for b in range(64):
            if plan["bucket_owner"][b] == dst:
                for i in range(int(counts[b])):
                    recs.append((rank, b, i))
    recs = torch.tensor(recs, dtype=torch.int64).reshape(-1, 3)
    send = pd.split_counts(counts, plan["bucket_owner"], world)
    recv = pd.exchange_counts(d, send)
    got = pd.exchange_records(d, recs, send, recv)
    ok = bool((torch.tensor([plan["bucket_owner"][int(b)] for b in got[:, 1]]) == rank).all()) if len(got) else True
    q.put((rank, t, c, plan["seeds"][rank], send, recv, int(got.shape[0]), ok, [int(x) for x in counts]))
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
    (r0, t0, c0, s0, send0, recv0, n0, ok0, cnt0), (r1, t1, c1, s1, send1, recv1, n1, ok1, cnt1) = res
    assert t0 == t1 == pytest.approx(0.020) and c0 == c1 == 2001          # max time, summed overlaps on every rank
    assert (s0, s1) == (1, 2)                                             # independent partitions, distinct seeds
    assert recv0 == [send0[0], send1[0]] and recv1 == [send0[1], send1[1]]
    assert n0 == sum(recv0) and n1 == sum(recv1) and ok0 and ok1
    assert n0 + n1 == sum(cnt0) + sum(cnt1)                               # nothing lost, nothing duplicated


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
