"""Oracle-free determinism check of the benched loop: the stream is periodic (5 batches), so batch b and batch b + 5 must
produce identical match lists and identical slots, in every mode.  A mismatch is a race (or an order-dependent reduction).
    python tools/gpu_determinism.py [steps]        env: URF_SP_TWO_STREAMS, URF_SINKHORN_RESIDENT, ..."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_pkg  # noqa: E402

U = load_pkg(); F, synth, P = U.frontend, U.synth, U.pipeline
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 26
spb, sgb = synth.pack_sp(synth.sp_weights(0)), synth.pack_sg(synth.sg_weights(0))
print(U._lib.lib().urf_build_info().decode(), "URF_SP_TWO_STREAMS =", os.environ.get("URF_SP_TWO_STREAMS", "(unset: off)"))
for (H, W) in ((376, 1241), (480, 640)):
    frames = synth.shift_stream(100, 40, H, W)
    dev = torch.device("cuda", 0)
    d_frames = torch.from_numpy(np.stack(frames)).to(dev)
    for prec in (2, 1):
        sp = F.SuperPoint(F.SuperPointConfig(max_keypoints=1000), max_height=H, max_width=W, max_batch=8, precision=prec)
        assert sp.build(spb)
        pms = []
        for _ in range(2):
            pm = F.PointMatching(F.SuperGlueConfig(image_width=640, image_height=512), max_pairs=8, precision=prec)
            assert pm.build(sgb)
            pms.append(pm)
        pipe = P.SlotRingPipeline(sp, pms, d_frames, 8, H, W, device=dev)
        pipe.prologue()
        lists, slots = {}, {}

        def rec(b, mt, res):
            lists[b] = [r.copy() for r in res]

        for b in range(steps):
            pipe.one_step(b, rec)
            if os.environ.get("URF_DET_SYNC") == "1":
                # snapshot of the slots SuperPoint(b + 1) has just been asked to write (synchronises: changes the timing)
                sp.sync()
                slots[b + 1] = pipe.ring[(b + 1) % 5].clone()
        pipe.drain(rec)
        bad_l = bad_s = 0
        for b in range(6, steps):
            for j in range(8):
                if not np.array_equal(lists[b][j], lists[b - 5][j]):
                    bad_l += 1
                    a, c = lists[b][j], lists[b - 5][j]
                    print(f"   {W}x{H} precision {prec}: batch {b} pair {j}: {len(a)} vs {len(c)} matches (batch {b - 5})")
        for b in range(6, steps + 1):
            if b in slots and b - 5 in slots and not torch.equal(slots[b], slots[b - 5]):
                d = (slots[b] != slots[b - 5]).any(dim=1).nonzero().flatten().tolist()
                bad_s += 1
                for j in d:
                    x, y = slots[b][j].cpu().numpy(), slots[b - 5][j].cpu().numpy()
                    k = np.nonzero(x != y)[0]
                    print(f"   {W}x{H} precision {prec}: slots of batch {b} frame {j} differ from batch {b - 5} in {k.size} words, first at {k[:4]}"
                          f" (header {x[:4].view(np.int32)} vs {y[:4].view(np.int32)})")
        print(f"{W}x{H} precision {prec}: {steps} steps, list mismatches {bad_l}, slot-batch mismatches {bad_s}, guard {sp.near_tie_reruns()}")
        del pipe, sp, pms
