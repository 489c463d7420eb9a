"""The exchange of include/urf.h through the C ABI on a real GPU (-m gpu): an RCCL communicator of ONE rank
(urf_comm_init with a unique id runs the same RCCL calls a world of 8 does), the all-gather of feature slots and the
gather of a matcher's device-resident results to the root, ordered on HIP streams.  The 8-GPU run is the driver's."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_rccl_allgather_and_gather_through_the_c_abi(U):
    import torch
    L = U._lib.lib()
    assert L.urf_device_count() >= 1
    D = U.dist
    slot_floats = L.urf_slot_bytes() // 4
    n = 3
    local = torch.arange(n * slot_floats, dtype=torch.float32, device="cuda").reshape(n, slot_floats) * 0.5
    stream = torch.cuda.Stream()
    for ident in (None, D.Comm.unique_id()):          # plain device copies / real RCCL, world of one
        comm = D.Comm(1, 0, 0, ident)
        assert L.urf_comm_world(comm._h) == 1 and L.urf_comm_rank(comm._h) == 0
        out = torch.zeros_like(local)
        torch.cuda.synchronize()
        comm.allgather_slots(local.data_ptr(), n, out.data_ptr(), stream.cuda_stream)
        root = torch.zeros(1000, dtype=torch.float32, device="cuda")
        comm.gather(local.data_ptr(), 4000, root.data_ptr(), 0, stream.cuda_stream)
        stream.synchronize()
        assert torch.equal(out, local) and torch.equal(root, local.reshape(-1)[:1000])
        del comm
    with pytest.raises(RuntimeError):
        D.Comm(2, 0, 0, None)                          # a world of two needs rank 0's unique id
    with pytest.raises(RuntimeError):
        D.Comm(2, 5, 0, D.Comm.unique_id())            # rank outside the world
