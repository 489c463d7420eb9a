"""Frame sharding and the feature-slot exchange for N GPUs of one node.

The reference is single-GPU; this is the data-parallel layer of
the new build: one process per GPU, frames of a batch block-distributed over the
ranks, SuperPoint locally, ONE all-gather (RCCL over xGMI when the backend is
"nccl"; gloo in the CPU tests) of fixed-size feature slots, then every rank
matches the pairs whose second frame it owns.  No other collective is on the
data path; the host tracker that consumes the match lists stays serial.
"""
import torch
import torch.distributed as dist


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
