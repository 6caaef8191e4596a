"""Probe (GPU box, 1 rank): (1) does a plain RCCL all_to_all_single move every byte of a large buffer?  — it does not above
2^30 bytes with RCCL 2.26.6; (2) does plass_amd.shard.TorchComm (chunked point-to-point pieces, self part copied) do it?"""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
from plass_amd.shard import TorchComm
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
comm = TorchComm(dist, torch.device("cuda:0"))
for nbytes in (960_000_000, 1_115_062_304, 1_400_000_000):
    n = nbytes // 8
    src = torch.arange(n, dtype=torch.int64, device="cuda")
    dst = torch.full((n,), -1, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    dist.all_to_all_single(dst, src, output_split_sizes=[n], input_split_sizes=[n])
    torch.cuda.synchronize()
    print("all_to_all_single  %d bytes: %d wrong words" % (nbytes, int((dst != src).sum().item())), flush=True)
    dst.fill_(-1)
    sb = (C.c_uint64 * 1)(n * 8)
    assert comm.struct.alltoallv_dev(None, src.data_ptr(), sb, dst.data_ptr(), sb) == 0, comm.error
    print("TorchComm.alltoallv %d bytes: %d wrong words" % (nbytes, int((dst != src).sum().item())), flush=True)
    dst.fill_(-1)
    assert comm.struct.allgatherv_dev(None, src.data_ptr(), n * 8, dst.data_ptr(), sb) == 0, comm.error
    print("TorchComm.allgatherv %d bytes: %d wrong words" % (nbytes, int((dst != src).sum().item())), flush=True)
    del dst, src
dist.destroy_process_group()
