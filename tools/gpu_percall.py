"""The per-call path of the unpatched reference caller (src/tracking.cc:338-377: SuperPoint::infer on one frame, then
PointMatching::MatchingPoints on two host feature matrices) through urf_sp_infer / urf_match, one pair at a time:
    python tools/gpu_percall.py [precision=3] [calls=40] [flag_every=0]
prints ms per call (median of 3 regions) and, with URF profiling on, the stage times of the last call.  Under
`rocprofv3 --kernel-trace --stats -- python3 tools/gpu_percall.py ...` the kernel statistics are those of this path alone."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as G  # noqa: E402

U = G.load_pkg()
F, synth = U.frontend, U.synth
prec = int(sys.argv[1]) if len(sys.argv) > 1 else 3
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 40
H, W = 480, 640
spb, sgb = synth.pack_sp(synth.sp_weights(0)), synth.pack_sg(synth.sg_weights(0))
fr = synth.shift_stream(100, 12, H, W)
sp = F.SuperPoint(F.SuperPointConfig(max_keypoints=1000), max_height=H, max_width=W, max_batch=1, precision=prec)
pm = F.PointMatching(F.SuperGlueConfig(image_width=640, image_height=512), max_pairs=1, precision=prec)
assert sp.build(spb) and pm.build(sgb)
feats = [sp.infer(f) for f in fr]
for j in range(9):                       # warm-up and the automatic guard calibration (8 pairs)
    pm.MatchingPoints(feats[j], feats[j + 1], True)
t_sp, t_pm = [], []
for _ in range(3):
    t0 = time.perf_counter()
    for i in range(calls):
        sp.infer(fr[i % 12])
    t_sp.append((time.perf_counter() - t0) / calls * 1e3)
    t0 = time.perf_counter()
    for i in range(calls):
        pm.MatchingPoints(feats[i % 11], feats[i % 11 + 1], True)
    t_pm.append((time.perf_counter() - t0) / calls * 1e3)
g = pm.near_tie_reruns()
print(f"precision {prec}: urf_sp_infer {np.median(t_sp):.3f} ms/frame, urf_match {np.median(t_pm):.3f} ms/pair "
      f"({1e3 / (np.median(t_sp) + np.median(t_pm)):.1f} frames/s one frame at a time); K = {feats[0].shape[0]}, "
      f"pairs redone {g['redone']} of {g['pairs']}")
F.set_profiling(True)
pm.MatchingPoints(feats[0], feats[1], True)
pm.MatchingPoints(feats[0], feats[1], True)
try:
    print("stage ms of one call (prep, kenc, gnn, final+score, sinkhorn, decode, ransac, attn, redo):", [round(v, 3) for v in pm.stage_ms()])
except Exception as e:  # noqa: BLE001
    print("stage_ms:", e)
