"""The exact redo chain alone (nothing else on the chip): stage times of an exact-mode handle on one and two pairs of the
bench stream, and the redo time of a strict-parity handle whose margin flags every pair.   python tools/gpu_redo_bench.py"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_pkg  # noqa: E402

U = load_pkg(); F, synth = U.frontend, U.synth
print(U._lib.lib().urf_build_info().decode())
H, W = 480, 640
spb, sgb = synth.pack_sp(synth.sp_weights(0)), synth.pack_sg(synth.sg_weights(0))
frames = synth.shift_stream(100, 4, H, W)
sp = F.SuperPoint(F.SuperPointConfig(max_keypoints=1000), max_height=H, max_width=W, max_batch=4)
assert sp.build(spb)
dev = torch.device("cuda", 0)
d = torch.from_numpy(np.stack(frames)).to(dev)
slots = torch.zeros((4, U._lib.lib().urf_slot_bytes() // 4), dtype=torch.float32, device=dev)
sp.infer_device(d.data_ptr(), 4, H, W, slots.data_ptr()); sp.sync()
F.set_profiling(True)
for P in (1, 2, 3):
    ex = F.PointMatching(F.SuperGlueConfig(image_width=640, image_height=512), max_pairs=P, precision=0)
    assert ex.build(sgb)
    s0 = [slots[j].data_ptr() for j in range(P)]; s1 = [slots[j + 1].data_ptr() for j in range(P)]
    for rep in range(3):
        t = time.perf_counter()
        ex.match_device_async(s0, s1, True); res = ex.fetch(P)
        dt = (time.perf_counter() - t) * 1e3
    st = ex.stage_ms()
    print(f"exact handle, {P} pair(s): wall {dt:.2f} ms; " + ", ".join(f"{n} {v:.3f}" for n, v in zip(F.PM_STAGES, st)))
    sx = F.PointMatching(F.SuperGlueConfig(image_width=640, image_height=512), max_pairs=8, precision=3, guard_margin=50.0, redo_flagged_pairs=2)
    assert sx.build(sgb)
    for rep in range(3):
        t = time.perf_counter()
        sx.match_device_async(s0, s1, True); res2 = sx.fetch(P)
        dt = (time.perf_counter() - t) * 1e3
    st = sx.stage_ms()
    assert res2 == res
    print(f"strict handle, {P} pair(s) all flagged: wall {dt:.2f} ms, fast pass {sum(st[:7]):.2f} ms, redo {st[8]:.2f} ms (same lists as the exact handle)")
