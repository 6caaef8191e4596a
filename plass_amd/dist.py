"""One-process-per-GPU plumbing for the hot path (torch.distributed; backend "nccl" = RCCL on ROCm, "gloo" in CPU tests).

bench.py's control plane: process-group set-up, the barrier-bracketed step time (max over ranks) and the overlap count
(sum over ranks).  The data path of a sharded run (one read set over the GPUs: DESIGN.md §6) is in plass_amd/shard.py;
`partition_plan` / `exchange_records` serve the older `--mode partitions` run (independent read sets, no data-path
collective) and the 2-rank gloo test of the count + record all-to-all.
"""
import os


def env_world():
    """(rank, local_rank, world_size) from the torch.distributed.run environment"""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init(backend, rank, world, device=None, timeout_s=None):
    import datetime
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    if dist.is_initialized():
        return dist
    kw = {}
    if device is not None:
        kw["device_id"] = device
    if timeout_s:
        kw["timeout"] = datetime.timedelta(seconds=timeout_s)      # a rank that died must not stall the others for 10 minutes
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return dist


def partition_plan(world, n_buckets=4096):
    """rank r owns the independent read partition seeded 1+r; k-mer hash bucket b (0..n_buckets-1, the top bits the
    hash-partition kernel already uses) is owned by rank b % world — the layout of exchange 1 in SURVEY.md §8e."""
    owner = [b % world for b in range(n_buckets)]
    return {"seeds": [1 + r for r in range(world)], "bucket_owner": owner}


def split_counts(bucket_counts, bucket_owner, world):
    """per-destination record counts for an all-to-all of bucketed records (send side)"""
    out = [0] * world
    for c, o in zip(bucket_counts, bucket_owner):
        out[o] += int(c)
    return out


def reduce_step(dist, elapsed_s, overlaps, device="cpu"):
    """max over ranks of the step time, sum over ranks of the overlap count (what bench.py reports)"""
    import torch
    if dist is None:
        return elapsed_s, overlaps
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    c = torch.tensor([overlaps], dtype=torch.int64, device=device)
    dist.all_reduce(c, op=dist.ReduceOp.SUM)
    return float(t.item()), int(c.item())


def exchange_counts(dist, send_counts, device="cpu"):
    """all-to-all of the per-destination counts: returns what every source will send to this rank"""
    import torch
    s = torch.tensor(send_counts, dtype=torch.int64, device=device)
    r = torch.empty_like(s)
    dist.all_to_all_single(r, s)
    return [int(x) for x in r.tolist()]


def exchange_records(dist, records, send_counts, recv_counts):
    """all-to-all(v) of fixed-size records (rows of a 2-D uint8/uint64 tensor), grouped by destination rank"""
    import torch
    out = torch.empty((sum(recv_counts),) + tuple(records.shape[1:]), dtype=records.dtype, device=records.device)
    dist.all_to_all_single(out, records, output_split_sizes=list(recv_counts), input_split_sizes=list(send_counts))
    return out
