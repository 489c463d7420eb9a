"""Frame sharding and the feature-slot exchange for N GPUs of one node.

The reference is single-GPU; this is the data-parallel layer of
the new build: one process per GPU, frames of a batch block-distributed over the
ranks, SuperPoint locally, ONE all-gather (RCCL over xGMI when the backend is
"nccl"; gloo in the CPU tests) of fixed-size feature slots, then every rank
matches the pairs whose second frame it owns.  No other collective is on the
data path; the host tracker that consumes the match lists stays serial.
"""
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from . import _lib
from ._lib import check


class Comm:
    """the exchange of include/urf.h (urf_comm_*): RCCL behind the C ABI.  One instance per rank.

        id_bytes = Comm.unique_id() on rank 0, shipped to the other ranks by any host channel;
        Comm(world, rank, device, id_bytes).  world == 1 with id_bytes=None never loads RCCL."""

    def __init__(self, world, rank, device, id_bytes=None):
        self.world, self.rank, self.device = int(world), int(rank), int(device)
        self._h = C.c_void_p()
        buf = None if id_bytes is None else C.create_string_buffer(bytes(id_bytes), 128)
        check(_lib.lib().urf_comm_init(self.world, self.rank, self.device, buf, C.byref(self._h)), "urf_comm_init")

    @classmethod
    def loopback(cls, world, device=0):
        """`world` logical ranks of ONE process on one device (urf_comm_init_loopback): a list of Comm, rank order"""
        hs = (C.c_void_p * world)()
        check(_lib.lib().urf_comm_init_loopback(int(world), int(device), hs), "urf_comm_init_loopback")
        out = []
        for r in range(world):
            c = cls.__new__(cls)
            c.world, c.rank, c.device, c._h = int(world), r, int(device), C.c_void_p(hs[r])
            out.append(c)
        return out

    @classmethod
    def init_all(cls, devices):
        """one process driving several devices (urf_comm_init_all = ncclCommInitAll): a list of Comm, rank i on devices[i].
        Bracket the collective calls made for them by one thread with Comm.group()."""
        n = len(devices)
        hs = (C.c_void_p * n)()
        devs = (C.c_int * n)(*[int(d) for d in devices])
        check(_lib.lib().urf_comm_init_all(n, devs, hs), "urf_comm_init_all")
        out = []
        for r in range(n):
            c = cls.__new__(cls)
            c.world, c.rank, c.device, c._h = n, r, int(devices[r]), C.c_void_p(hs[r])
            out.append(c)
        return out

    class group:
        """with Comm.group(): ... -- ncclGroupStart / ncclGroupEnd around the calls one thread makes for several ranks"""

        def __enter__(self):
            check(_lib.lib().urf_comm_group_start(), "urf_comm_group_start")

        def __exit__(self, *a):
            check(_lib.lib().urf_comm_group_end(), "urf_comm_group_end")

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(128)
        check(_lib.lib().urf_comm_unique_id(buf), "urf_comm_unique_id")
        return buf.raw

    def allgather_slots(self, d_local, nslots, d_all, stream=None):
        """d_all[world * nslots] <- every rank's d_local[nslots] (device pointers), enqueued on `stream` (hipStream_t)"""
        check(_lib.lib().urf_comm_allgather_slots(self._h, C.c_void_p(d_local), int(nslots), C.c_void_p(d_all),
                                                  C.c_void_p(stream)), "urf_comm_allgather_slots")

    def gather(self, d_send, nbytes, d_recv, root=0, stream=None):
        check(_lib.lib().urf_comm_gather(self._h, C.c_void_p(d_send), C.c_size_t(nbytes), C.c_void_p(d_recv), int(root),
                                         C.c_void_p(stream)), "urf_comm_gather")

    def __del__(self):
        try:
            if self._h:
                _lib.lib().urf_comm_destroy(self._h)
                self._h = None
        except Exception:
            pass


def plan_pairs(world, rank, per_rank):
    """urf_comm_plan_pairs: (first, second) indices into the gathered slots of the pairs `rank` matches in a step;
    first == -1 is the carried last frame of the previous step"""
    a = np.zeros(per_rank, np.int32)
    b = np.zeros(per_rank, np.int32)
    check(_lib.lib().urf_comm_plan_pairs(int(world), int(rank), int(per_rank), a.ctypes.data_as(C.c_void_p),
                                         b.ctypes.data_as(C.c_void_p)), "urf_comm_plan_pairs")
    return a, b


def shard_range(n_frames, rank, world):
    """block-contiguous shard [lo, hi) of a batch of n_frames (n_frames % world == 0)."""
    assert n_frames % world == 0, "batch must divide evenly over the ranks"
    per = n_frames // world
    return rank * per, (rank + 1) * per


def pairs_for_rank(n_frames, rank, world):
    """pairs (prev, cur) over GLOBAL frame indices owned by `rank`: cur in its
    shard, prev = cur-1; prev == -1 denotes the carried last frame of the
    previous batch."""
    lo, hi = shard_range(n_frames, rank, world)
    return [(t - 1, t) for t in range(lo, hi)]


def all_gather_slots(local_slots, world):
    """local_slots: [per, slot_floats] tensor on this rank's device.  Returns the
    [world*per, slot_floats] tensor holding every rank's slots in global frame
    order.  world == 1 is a no-op (no process group needed)."""
    if world == 1:
        return local_slots
    out = torch.empty((world * local_slots.shape[0],) + tuple(local_slots.shape[1:]),
                      dtype=local_slots.dtype, device=local_slots.device)
    if dist.get_backend() == "gloo" and local_slots.is_cuda:
        # test rig only (several ranks sharing one GPU): stage through the host
        tmp = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(tmp, local_slots.contiguous().cpu())
        out.copy_(tmp)
        return out
    dist.all_gather_into_tensor(out, local_slots.contiguous())
    return out


def max_over_ranks(value, device, world):
    if world == 1:
        return value
    if dist.get_backend() == "gloo":
        device = torch.device("cpu")
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
