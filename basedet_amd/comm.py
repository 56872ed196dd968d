"""Data-parallel transport of the HIP training path: the bd_comm_* entry points of libbasedet_hip.so (RCCL over xGMI).

One process per GPU.  The reference's `dist.*` calls on the path map as follows
  dist.bcast_list_(params / buffers)        configs/detection_cfg.py:80-82        -> Comm.bcast
  dist.make_allreduce_cb(reduce_mode)       solver/default_solver.py:58-63,121    -> Comm.allreduce_async + Comm.wait
  all_reduce_mean(num_fg / sum_ctr)         models/det/fcos.py:143-144            -> Comm.allreduce(..., "avg")
`torch.distributed` is used for ONE thing: its TCP store carries the 128-byte RCCL id from rank 0 to the other ranks
(env:// rendezvous: RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT, also under torch.distributed.run's agent store).
No c10d process group is created on the GPU path.

`GlooComm` is the CPU stand-in with the same interface over an initialised gloo process group: host-logic tests only
(tests/test_dist_cpu.py) -- there are no kernels on a CPU, so nothing of the product path runs through it.
"""
import ctypes as C
import os

import torch

from . import _lib
from ._lib import check, ptr

_DTYPES = {torch.float32: 0, torch.bfloat16: 1, torch.int32: 2, torch.float64: 3}
_OPS = {"sum": 0, "max": 1, "avg": 2}

_active = None


def get_comm():
    """The process's communicator, or None on a single-process run."""
    return _active


def set_comm(comm):
    global _active
    _active = comm
    return comm


def world_size():
    return _active.world if _active is not None else 1


def rank():
    return _active.rank if _active is not None else 0


def _exchange_id(uid, rank_, world, key):
    """rank 0 publishes the RCCL id, the others fetch it (torch.distributed's TCP store is the only c10d piece in use)."""
    import torch.distributed as dist
    store = None
    try:
        from torch.distributed import rendezvous
        store, _, _ = next(rendezvous("env://", rank_, world))       # honours TORCHELASTIC_USE_AGENT_STORE
    except Exception:
        store = dist.TCPStore(os.environ.get("MASTER_ADDR", "127.0.0.1"), int(os.environ["MASTER_PORT"]), world, rank_ == 0)
    if rank_ == 0:
        store.set(key, bytes(uid))
        return bytes(uid), store
    return bytes(store.get(key)), store


class Comm:
    """bd_comm_t wrapper.  Tensors are passed as device pointers; every call is stream-ordered and returns at once."""

    def __init__(self, rank_=0, world=1, device=0, key="bd_comm_id/0"):
        lib = _lib.load()
        uid = (C.c_char * 128)()
        self._store = None
        if rank_ == 0:
            check(lib.bd_comm_unique_id(uid), "bd_comm_unique_id")
        if world > 1:
            raw, self._store = _exchange_id(uid, rank_, world, key)
            uid = (C.c_char * 128).from_buffer_copy(raw)
        self._h = C.c_void_p()
        check(lib.bd_comm_init(C.byref(self._h), uid, rank_, world, device), "bd_comm_init")
        self.rank, self.world, self.device = rank_, world, device
        self._lib = lib
        self.stream = torch.cuda.ExternalStream(int(lib.bd_comm_stream(self._h)), device=torch.device("cuda", device))

    @classmethod
    def from_env(cls):
        """RANK / LOCAL_RANK / WORLD_SIZE as exported by the launcher (bench.py's own, or torch.distributed.run)."""
        r, w = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
        dev = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(dev)
        return cls(r, w, dev, key="bd_comm_id/" + os.environ.get("TORCHELASTIC_RESTART_COUNT", "0"))

    @staticmethod
    def _sp(stream):
        return C.c_void_p((stream or torch.cuda.current_stream()).cuda_stream)

    def bcast(self, t, root=0, stream=None):
        check(self._lib.bd_comm_bcast(self._h, ptr(t), t.numel(), _DTYPES[t.dtype], root, self._sp(stream)), "bd_comm_bcast")

    def allreduce(self, t, op="sum", stream=None):
        """In place, ordered on the caller's stream (forward-side exchanges of a few scalars)."""
        check(self._lib.bd_comm_allreduce(self._h, ptr(t), t.numel(), _DTYPES[t.dtype], _OPS[op], self._sp(stream)), "bd_comm_allreduce")

    def allreduce_async(self, t, producers, op="sum"):
        """In place on the communication stream, after everything enqueued so far on the `producers` streams."""
        arr = (C.c_void_p * len(producers))(*[s.cuda_stream for s in producers])
        check(self._lib.bd_comm_allreduce_async(self._h, ptr(t), t.numel(), _DTYPES[t.dtype], _OPS[op], arr, len(producers)),
              "bd_comm_allreduce_async")

    def allreduce_async_bf16(self, t, tmp, producers, op="sum"):
        """As allreduce_async, with the fp32 tensor compressed to bf16 on the wire (tmp: bf16 scratch of t.numel() elements)."""
        assert t.dtype == torch.float32 and tmp.dtype == torch.bfloat16 and tmp.numel() >= t.numel()
        arr = (C.c_void_p * len(producers))(*[s.cuda_stream for s in producers])
        check(self._lib.bd_comm_allreduce_async_bf16(self._h, ptr(t), ptr(tmp), t.numel(), _OPS[op], arr, len(producers)),
              "bd_comm_allreduce_async_bf16")

    def wait(self, stream=None):
        check(self._lib.bd_comm_wait(self._h, self._sp(stream)), "bd_comm_wait")

    def barrier(self):
        """Host barrier: a one-element all-reduce, then a device sync."""
        if not hasattr(self, "_tok"):
            self._tok = torch.zeros(1, dtype=torch.float32, device=torch.device("cuda", self.device))
        self.allreduce(self._tok)
        torch.cuda.synchronize()

    def destroy(self):
        if self._h:
            self._lib.bd_comm_destroy(self._h)
            self._h = C.c_void_p()


class GlooComm:
    """Same interface over an initialised torch.distributed (gloo) group and CPU tensors: CPU tests of the bucket logic."""

    stream = None

    def __init__(self):
        import torch.distributed as dist
        self._d = dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()

    def bcast(self, t, root=0, stream=None):
        self._d.broadcast(t, src=root)

    def allreduce(self, t, op="sum", stream=None):
        d = self._d
        d.all_reduce(t, op={"sum": d.ReduceOp.SUM, "max": d.ReduceOp.MAX, "avg": d.ReduceOp.SUM}[op])
        if op == "avg":
            t.div_(self.world)

    def allreduce_async(self, t, producers, op="sum"):
        self.allreduce(t, op)

    def allreduce_async_bf16(self, t, tmp, producers, op="sum"):
        """CPU stand-in of the compressed exchange: every rank rounds its contribution to bf16, the sum is rounded to bf16 again."""
        tmp = tmp[: t.numel()].view_as(t)
        tmp.copy_(t)                                     # fp32 -> bf16 (round to nearest even)
        r = tmp.float()
        self.allreduce(r, op)
        t.copy_(r.to(torch.bfloat16).float())

    def wait(self, stream=None):
        pass

    def barrier(self):
        self._d.barrier()

    def destroy(self):
        pass
