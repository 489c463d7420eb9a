"""One rank of a REAL RCCL world (tests/test_gpu_comm.py, boxes with >= 2 GPUs): its shard of the bench stream through the
step loop bench.py runs (ur-mvo_amd/pipeline.py) with urf_comm_init(world, rank, device = rank, id); everything it fetched,
and on rank 0 what the gathers delivered, goes to an .npz.     python rccl_worker.py RANK WORLD IDFILE OUT H W STEPS PREC"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_pkg  # noqa: E402


def main():
    rank, world, idfile, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    H, W, steps, prec = (int(v) for v in sys.argv[5:9])
    import torch
    U = load_pkg()
    F, D, P, synth = U.frontend, U.dist, U.pipeline, U.synth
    B, M = 4, 2
    NB = M + 3
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    if rank == 0:
        ident = D.Comm.unique_id()
        with open(idfile + ".tmp", "wb") as f:
            f.write(ident)
        os.replace(idfile + ".tmp", idfile)
    else:
        t0 = time.time()
        while not os.path.exists(idfile):
            assert time.time() - t0 < 120, "rank 0 never wrote the RCCL id"
            time.sleep(0.05)
        ident = open(idfile, "rb").read()
    comm = D.Comm(world, rank, rank, ident)
    spb, sgb = synth.pack_sp(synth.sp_weights(0)), synth.pack_sg(synth.sg_weights(0))
    frames = synth.shift_stream(100, 40, H, W)
    n = len(frames)
    sp = F.SuperPoint(F.SuperPointConfig(max_keypoints=1000), max_height=H, max_width=W, max_batch=B, device=rank, precision=prec)
    assert sp.build(spb)
    pms = []
    for _ in range(M):
        pm = F.PointMatching(F.SuperGlueConfig(image_width=640, image_height=512), max_pairs=B, device=rank, precision=prec)
        assert pm.build(sgb)
        pms.append(pm)
    idx = [((k * world + rank) * B + j) % n for k in range(NB) for j in range(B)]
    d_frames = torch.from_numpy(np.stack([frames[i] for i in idx])).to(dev)
    pipe = P.SlotRingPipeline(sp, pms, d_frames, B, H, W, device=dev, rank=rank, world=world, comm=comm, keep_gathered=(rank == 0))
    pipe.prologue()
    fetched = dict(pipe.run(0, steps) + pipe.drain())
    pipe.finish_exchange(steps - 1)
    sp.sync()
    torch.cuda.synchronize()
    cnt = np.zeros((steps, B), np.int32)
    mt = np.zeros((steps, B, 1024), F.MATCH_DTYPE)
    for b in range(steps):
        for j in range(B):
            m = fetched[b][j]
            cnt[b, j] = len(m)
            mt[b, j, :len(m)] = m
    res = {"counts": cnt, "matches": mt.view(np.int32).reshape(steps, B, 1024, 3),
           "gathered": np.stack([pipe.gathered_buf[k].cpu().numpy() for k in range(steps)]),
           "comm_world": np.array([U._lib.lib().urf_comm_world(comm._h), U._lib.lib().urf_comm_rank(comm._h)])}
    if rank == 0:
        res["root_counts"] = np.stack([pipe.gather_log[b][0] for b in range(steps)])
        res["root_matches"] = np.stack([pipe.gather_log[b][1].view(np.int32).reshape(world, B, 1024, 3) for b in range(steps)])
    np.savez(out, **res)


if __name__ == "__main__":
    main()
