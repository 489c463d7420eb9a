"""Where the time of a per-call MatchingPoints goes when its pair is flagged (strict mode, every pair flagged by an absurd margin):
host time of the enqueue of the fast pass, of fetch_begin (wait for the fast pass + staging + enqueue of the exact pass) and of
fetch_end (wait for the exact pass + hand-out), against the GPU times of the two passes.    python tools/gpu_flagged_call.py"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_pkg  # noqa: E402

U = load_pkg(); F, synth = U.frontend, U.synth
H, W = 480, 640
spb, sgb = synth.pack_sp(synth.sp_weights(0)), synth.pack_sg(synth.sg_weights(0))
frames = synth.shift_stream(100, 3, H, W)
sp = F.SuperPoint(F.SuperPointConfig(max_keypoints=1000), max_height=H, max_width=W, max_batch=3)
assert sp.build(spb)
dev = torch.device("cuda", 0)
d = torch.from_numpy(np.stack(frames)).to(dev)
slots = torch.zeros((3, U._lib.lib().urf_slot_bytes() // 4), dtype=torch.float32, device=dev)
sp.infer_device(d.data_ptr(), 3, H, W, slots.data_ptr()); sp.sync()
feats = [F.slot_to_host(slots[j].data_ptr()) for j in range(3)]
F.set_profiling(True)
for P in (1, 2):
    sx = F.PointMatching(F.SuperGlueConfig(image_width=640, image_height=512), max_pairs=P, precision=3, guard_margin=50.0,
                         calibrate_pairs=-1, redo_flagged_pairs=2, audit_period=-1)
    assert sx.build(sgb)
    s0 = [slots[j].data_ptr() for j in range(P)]; s1 = [slots[j + 1].data_ptr() for j in range(P)]
    acc = []
    first = None
    for rep in range(6):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); sx.match_device_async(s0, s1, True)
        t1 = time.perf_counter(); sx.fetch_begin(P)
        t2 = time.perf_counter(); sx.fetch_end(P, as_arrays=True)
        t3 = time.perf_counter()
        if rep == 0:
            first = (t3 - t0) * 1e3
        if rep >= 2:
            acc.append(((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, sum(sx.stage_ms()[:7]), sx.stage_ms()[8] if len(sx.stage_ms()) > 8 else -1))
    a = np.mean(np.array(acc), axis=0)
    print(f"{P} pair(s), device batch: enqueue of the fast pass {a[0]:.2f} ms, fetch_begin {a[1]:.2f} ms, fetch_end {a[2]:.2f} ms (host); GPU: fast pass {a[3]:.2f} ms, redo {a[4]:.2f} ms; the handle's FIRST flagged batch took {first:.2f} ms in all")
    if P == 1:
        acc = []
        for rep in range(6):
            t0 = time.perf_counter(); sx.MatchingPoints(feats[0], feats[1], True); acc.append((time.perf_counter() - t0) * 1e3)
        print(f"1 pair, host call urf_match (flagged): {np.mean(acc[2:]):.2f} ms")
