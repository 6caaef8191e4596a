"""One-process-per-GPU plumbing for the hot path (torch.distributed; backend "nccl" = RCCL on ROCm, "gloo" in CPU tests).

bench.py's control plane: process-group set-up, the barrier-bracketed step time (max over ranks) and the overlap count
(sum over ranks).  The data path of a sharded run (one read set over the GPUs: DESIGN.md §6) is in the library
(plass_amd/csrc/comm.hip, comm_rccl.hip) and, for the caller-supplied collectives, in plass_amd/shard.py.
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
