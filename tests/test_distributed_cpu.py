"""world_size-2 gloo test (CPU) of the N>1 path: block sharding, the all-gather
of feature slots and the pair ownership, with the slot layout of the product."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import load_pkg
    U = load_pkg()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    per, slot = 4, 64  # small stand-in slots: header int + payload
    lo, hi = U.dist.shard_range(per * world, rank, world)
    local = torch.zeros((per, slot), dtype=torch.float32)
    for j in range(per):
        local[j, 0] = float(lo + j)          # "K" header carries the global frame id
        local[j, 1:] = float(rank) + 0.5
    allslots = U.dist.all_gather_slots(local, world)
    ok = allslots.shape == (per * world, slot)
    ok &= bool((allslots[:, 0] == torch.arange(per * world, dtype=torch.float32)).all())
    ok &= bool((allslots[lo:hi] == local).all())
    pairs = U.dist.pairs_for_rank(per * world, rank, world)
    ok &= pairs == [(t - 1, t) for t in range(lo, hi)]
    mx = U.dist.max_over_ranks(float(rank + 1), torch.device("cpu"), world)
    ok &= mx == float(world)
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_all_gather_slots_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 500)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]


def test_world1_is_a_noop():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import load_pkg
    U = load_pkg()
    t = torch.arange(12, dtype=torch.float32).reshape(3, 4)
    assert U.dist.all_gather_slots(t, 1) is t
    assert U.dist.max_over_ranks(3.5, torch.device("cpu"), 1) == 3.5


def test_pair_ownership_plan_of_the_c_abi():
    """urf_comm_plan_pairs (host-only, no process group): over all ranks every frame of a step is the second frame of
    exactly one pair, each pair's first frame is its predecessor in global order, the only carried frame (-1) is the
    predecessor of the step's first frame -- and the plan equals the Python sharding helpers used by the gloo rig"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import load_pkg
    U = load_pkg()
    for world, per in [(1, 8), (2, 8), (8, 4), (8, 8), (3, 5)]:
        seconds = []
        for rank in range(world):
            a, b = U.dist.plan_pairs(world, rank, per)
            lo, hi = U.dist.shard_range(per * world, rank, world)
            assert list(b) == list(range(lo, hi)) and list(a) == [t - 1 for t in range(lo, hi)]
            assert [(int(x), int(y)) for x, y in zip(a, b)] == U.dist.pairs_for_rank(per * world, rank, world)
            seconds += list(b)
        assert sorted(seconds) == list(range(world * per))
    a, _ = U.dist.plan_pairs(4, 0, 4)
    assert a[0] == -1 and (U.dist.plan_pairs(4, 1, 4)[0] >= 0).all()
    import pytest
    with pytest.raises(RuntimeError):
        U.dist.plan_pairs(2, 2, 4)
