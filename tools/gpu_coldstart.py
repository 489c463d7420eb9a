"""Cold-start repeatability of the guarded loop: fresh handles, 8 steps, the lists of every batch compared across repetitions
(the first must equal all others).    python tools/gpu_coldstart.py [repetitions] [H W]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_pkg  # noqa: E402

U = load_pkg(); F, synth, P = U.frontend, U.synth, U.pipeline
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
H, W = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (376, 1241)
prec = int(os.environ.get("URF_PREC", "2"))
spb, sgb = synth.pack_sp(synth.sp_weights(0)), synth.pack_sg(synth.sg_weights(0))
frames = synth.shift_stream(100, 40, H, W)
dev = torch.device("cuda", 0)
d_frames = torch.from_numpy(np.stack(frames)).to(dev)
ref = None
for rep in range(reps):
    sp = F.SuperPoint(F.SuperPointConfig(max_keypoints=1000), max_height=H, max_width=W, max_batch=8, precision=prec)
    assert sp.build(spb)
    pms = []
    for _ in range(2):
        pm = F.PointMatching(F.SuperGlueConfig(image_width=640, image_height=512), max_pairs=8, precision=prec)
        assert pm.build(sgb)
        pms.append(pm)
    pipe = P.SlotRingPipeline(sp, pms, d_frames, 8, H, W, device=dev)
    pipe.prologue()
    got = dict(pipe.run(0, 7) + pipe.drain())
    sp.sync()
    ring = pipe.ring.clone()
    cur = {b: [r.copy() for r in got[b]] for b in got}
    if ref is None:
        ref, ref_ring = cur, ring
    else:
        bad = [(b, j, len(cur[b][j]), len(ref[b][j])) for b in cur for j in range(8) if not np.array_equal(cur[b][j], ref[b][j])]
        sd = (ring != ref_ring).any(dim=2).nonzero().tolist()
        print(f"repetition {rep}: lists differing from repetition 0: {bad}; slots differing: {sd}; fallbacks {[m.sinkhorn_fallbacks() for m in pms]}", flush=True)
    del pipe, sp, pms
print("done")
