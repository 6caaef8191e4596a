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
