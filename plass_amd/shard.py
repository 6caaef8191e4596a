"""Communicators for ONE read set sharded over several GPUs (include/plasship.h: plasship_comm / plasship_ctx_set_comm).

The library exchanges device buffers through three collectives its caller supplies:

* `TorchComm`  — torch.distributed (backend "nccl" = RCCL over xGMI): what bench.py uses, one process per GPU.
* `LocalGroup` — an in-process implementation for tests: W contexts on ONE GPU driven by W Python threads; the
  "exchange" is a device-to-device copy.  It runs exactly the code a multi-GPU run runs (partition by owner,
  all-to-all(v), halo of the run scan, all-gather of the extended sequences) on a 1-GPU box.

The reference's counterpart is the MPI split of kmermatcher by k-mer hash range
(lib/mmseqs/src/linclust/kmermatcher.cpp:312,736-778), merged through files.
"""
import ctypes as C
import threading

from . import _lib

_AG_HOST = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64)
_A2A_DEV = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), C.c_void_p, C.POINTER(C.c_uint64))
_AGV_DEV = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.POINTER(C.c_uint64))


class CommStruct(C.Structure):
    """struct plasship_comm"""
    _fields_ = [("rank", C.c_int), ("world", C.c_int), ("user", C.c_void_p), ("allgather_host", _AG_HOST),
                ("alltoallv_dev", _A2A_DEV), ("allgatherv_dev", _AGV_DEV), ("stream_ordered", C.c_int)]


def owned_range(n, rank, world):
    """ids [lo, hi) of the queries / representatives rank owns: lo = ceil(rank * n / world)"""
    return (rank * n + world - 1) // world, ((rank + 1) * n + world - 1) // world


class _CommBase:
    """keeps the ctypes callbacks alive and installs them on a context"""

    def __init__(self, rank, world):
        self.rank, self.world = rank, world
        self.error = None
        self.bytes_moved, self.seconds, self.calls = 0, 0.0, 0      # device bytes this rank sent, time inside collectives, calls
        self._cbs = (_AG_HOST(self._wrap(self._allgather_host)), _A2A_DEV(self._wrap(self._alltoallv_dev)),
                     _AGV_DEV(self._wrap(self._allgatherv_dev)))
        self.struct = CommStruct(rank, world, None, *self._cbs, 0)

    def _wrap(self, fn):
        def cb(user, *a):
            import time
            t0 = time.perf_counter()
            try:
                fn(*a)
                self.seconds += time.perf_counter() - t0
                self.calls += 1
                return 0
            except BaseException as e:      # never let an exception cross the C boundary
                self.error = e
                self._abort()
                return 1
        return cb

    def _abort(self):
        pass

    def install(self, ctx):
        lib = ctx.lib
        _lib._check(lib.plasship_ctx_set_comm(ctx.h, C.byref(self.struct)), "plasship_ctx_set_comm")
        ctx._comm = self            # keep alive as long as the context

    @staticmethod
    def uninstall(ctx):
        _lib._check(ctx.lib.plasship_ctx_set_comm(ctx.h, None), "plasship_ctx_set_comm")
        ctx._comm = None


# ---------------------------------------------------------------------------------------------------
# in-process group: W ranks = W threads, W contexts on one GPU
# ---------------------------------------------------------------------------------------------------
class LocalGroup:
    def __init__(self, world):
        self.world = world
        self.barrier = threading.Barrier(world)
        self.slots = [None] * world
        self.comms = [LocalComm(self, r) for r in range(world)]

    def run(self, fn, contexts):
        """fn(rank, ctx) on every rank concurrently (communicators installed); returns the list of results"""
        out = [None] * self.world
        err = [None] * self.world

        def work(r):
            try:
                self.comms[r].ctx = contexts[r]
                self.comms[r].install(contexts[r])
                out[r] = fn(r, contexts[r])
            except BaseException as e:
                err[r] = e
                self.barrier.abort()
            finally:
                try:
                    _CommBase.uninstall(contexts[r])
                except Exception:
                    pass

        th = [threading.Thread(target=work, args=(r,)) for r in range(self.world)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        for r in range(self.world):
            e = err[r] or self.comms[r].error
            if e is not None and not isinstance(e, threading.BrokenBarrierError):
                raise e
        for e in err:
            if e is not None:
                raise e
        return out


class LocalComm(_CommBase):
    def __init__(self, group, rank):
        super().__init__(rank, group.world)
        self.g = group
        self.ctx = None

    def _abort(self):
        self.g.barrier.abort()

    def _copy(self, dst, src, n):
        _lib._check(self.ctx.lib.plasship_ctx_copy_d2d(self.ctx.h, C.c_void_p(dst), C.c_void_p(src), n), "plasship_ctx_copy_d2d")

    def _allgather_host(self, send, recv, nbytes):
        g = self.g
        g.slots[self.rank] = C.string_at(send, nbytes)
        g.barrier.wait()
        for r in range(self.world):
            C.memmove(recv + r * nbytes, g.slots[r], nbytes)
        g.barrier.wait()

    def _alltoallv_dev(self, d_send, send_bytes, d_recv, recv_bytes):
        g, W = self.g, self.world
        g.slots[self.rank] = (d_send or 0, [int(send_bytes[i]) for i in range(W)])
        g.barrier.wait()
        roff = 0
        for r in range(W):
            src, sb = g.slots[r]
            n = sb[self.rank]
            if n != int(recv_bytes[r]):
                raise RuntimeError("all-to-all size mismatch")
            if n:
                self._copy(d_recv + roff, src + sum(sb[:self.rank]), n)
            roff += n
        g.barrier.wait()           # nobody reuses its send buffer before everybody has read it

    def _allgatherv_dev(self, d_send, nbytes, d_recv, recv_bytes):
        g, W = self.g, self.world
        g.slots[self.rank] = (d_send or 0, int(nbytes))
        g.barrier.wait()
        roff = 0
        for r in range(W):
            src, n = g.slots[r]
            if n != int(recv_bytes[r]):
                raise RuntimeError("all-gather size mismatch")
            if n:
                self._copy(d_recv + roff, src, n)
            roff += n
        g.barrier.wait()


# ---------------------------------------------------------------------------------------------------
# native RCCL communicator (include/plasship_rccl.h): the C++ side owns the collectives, Python only distributes the id
# ---------------------------------------------------------------------------------------------------
RCCL_ID_BYTES = 128


def rccl_unique_id():
    """ncclGetUniqueId through the library (rank 0 calls this; every rank must receive the same 128 bytes)"""
    lib = _lib.load_library()
    buf = C.create_string_buffer(RCCL_ID_BYTES)
    _lib._check(lib.plasship_rccl_get_unique_id(buf), "plasship_rccl_get_unique_id")
    return buf.raw


class RcclComm:
    """plasship_rccl_comm: ncclSend / ncclRecv groups on the context's stream; installs itself on the context (collective)"""

    def __init__(self, ctx, rank, world, unique_id):
        self.ctx, self.rank, self.world = ctx, rank, world
        self.h = C.c_void_p()
        _lib._check(ctx.lib.plasship_rccl_comm_create(ctx.h, rank, world, C.c_char_p(bytes(unique_id)), C.byref(self.h)), "plasship_rccl_comm_create")

    def stats(self, reset=False):
        b = C.c_uint64(); s = C.c_double(); n = C.c_uint64()
        _lib._check(self.ctx.lib.plasship_rccl_comm_stats(self.h, C.byref(b), C.byref(s), C.byref(n), int(reset)), "plasship_rccl_comm_stats")
        return b.value, s.value, n.value

    def destroy(self):
        if self.h:
            self.ctx.lib.plasship_rccl_comm_destroy(self.ctx.h, self.h); self.h = C.c_void_p()


# ---------------------------------------------------------------------------------------------------
# torch.distributed (RCCL): one process per GPU
# ---------------------------------------------------------------------------------------------------
class _DevPtr:
    """zero-copy view of library-owned device memory for torch (CUDA array interface v2)"""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


class TorchComm(_CommBase):
    def _abort(self):
        # a collective failed on this rank: tear the group down so that the peers' pending operations fail instead of waiting
        try:
            if self.dist.is_initialized():
                self.dist.destroy_process_group(self.group)
        except Exception:
            pass

    """collectives over a torch.distributed process group.  device: torch.device of this rank's GPU; None: the "device"
    pointers are host pointers and the tensors CPU tensors (the gloo tests drive the same split arithmetic that way)."""

    def __init__(self, dist, device=None, group=None):
        super().__init__(dist.get_rank(group), dist.get_world_size(group))
        self.dist, self.device, self.group = dist, device, group

    def _view(self, ptr, nbytes):
        import torch
        if nbytes == 0:
            return torch.empty(0, dtype=torch.uint8, device=self.device)
        if self.device is None:
            return torch.frombuffer((C.c_uint8 * nbytes).from_address(ptr), dtype=torch.uint8)
        return torch.as_tensor(_DevPtr(ptr, nbytes), device=self.device)

    def _sync(self):
        if self.device is not None:
            import torch
            torch.cuda.synchronize(self.device)

    def _allgather_host(self, send, recv, nbytes):
        import torch
        t = torch.frombuffer(bytearray(C.string_at(send, nbytes)), dtype=torch.uint8)
        if self.device is not None:
            t = t.to(self.device)
        out = torch.empty(nbytes * self.world, dtype=torch.uint8, device=t.device)
        self.dist.all_gather_into_tensor(out, t, group=self.group)
        host = out.cpu().numpy().tobytes()
        C.memmove(recv, host, len(host))

    # Device data moves as point-to-point messages (ncclSend / ncclRecv pairs in one group per round — what an
    # all-to-all is underneath), in pieces of at most CHUNK bytes: RCCL 2.26.6 silently drops half of a message of more
    # than 2^30 bytes (tools/rccl_a2a_probe.py shows it with a plain all_to_all_single), and a k-mer record exchange at
    # 1 M reads per GPU is already 1 GB.  Both ends of a pair know the message size, so they agree on the pieces
    # without talking; the piece a rank keeps for itself is a plain device copy.
    CHUNK = 256 << 20

    def _exchange(self, inp, out, soff, sb, roff, rb):
        """inp[soff[r] : soff[r] + sb[r]] -> rank r;  out[roff[r] : roff[r] + rb[r]] <- rank r   (uint8 tensors)"""
        dist, W, me, CH = self.dist, self.world, self.rank, self.CHUNK
        if sb[me]:
            out[roff[me]:roff[me] + rb[me]].copy_(inp[soff[me]:soff[me] + sb[me]])
        rounds = max([0] + [(max(sb[r], rb[r]) + CH - 1) // CH for r in range(W) if r != me])
        for k in range(rounds):
            ops = []
            for r in range(W):
                if r == me:
                    continue
                lo, hi = min(k * CH, sb[r]), min((k + 1) * CH, sb[r])
                if hi > lo:
                    ops.append(dist.P2POp(dist.isend, inp[soff[r] + lo:soff[r] + hi], r if self.group is None else dist.get_global_rank(self.group, r), self.group))
                lo, hi = min(k * CH, rb[r]), min((k + 1) * CH, rb[r])
                if hi > lo:
                    ops.append(dist.P2POp(dist.irecv, out[roff[r] + lo:roff[r] + hi], r if self.group is None else dist.get_global_rank(self.group, r), self.group))
            if ops:
                for q in dist.batch_isend_irecv(ops):
                    q.wait()
        self._sync()

    def _alltoallv_dev(self, d_send, send_bytes, d_recv, recv_bytes):
        W = self.world
        sb = [int(send_bytes[i]) for i in range(W)]
        rb = [int(recv_bytes[i]) for i in range(W)]
        soff = [sum(sb[:r]) for r in range(W)]
        roff = [sum(rb[:r]) for r in range(W)]
        self._exchange(self._view(d_send, sum(sb)), self._view(d_recv, sum(rb)), soff, sb, roff, rb)
        self.bytes_moved += sum(sb)

    def _allgatherv_dev(self, d_send, nbytes, d_recv, recv_bytes):
        W = self.world
        rb = [int(recv_bytes[i]) for i in range(W)]
        nbytes = int(nbytes)
        if rb[self.rank] != nbytes:
            raise RuntimeError("all-gather size mismatch")
        roff = [sum(rb[:r]) for r in range(W)]
        self._exchange(self._view(d_send, nbytes), self._view(d_recv, sum(rb)), [0] * W, [nbytes] * W, roff, rb)
        self.bytes_moved += nbytes * max(W - 1, 1)
