"""Where do the fast and the exact mode disagree on the bench stream?  Per frame: keypoint sets; per pair: match
sets (as pixel correspondences) with and without the RANSAC stage.  python tools/gpu_fastdiff.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_pkg  # noqa: E402

U = load_pkg(); F, synth = U.frontend, U.synth
H, W, N = 480, 640, 25
spb, sgb = synth.pack_sp(synth.sp_weights(0)), synth.pack_sg(synth.sg_weights(0))
frames = synth.shift_stream(100, N, H, W)
feats = {}
for prec in (0, 1):
    sp = F.SuperPoint(F.SuperPointConfig(max_keypoints=1000), max_height=H, max_width=W, precision=prec)
    assert sp.build(spb)
    feats[prec] = [sp.infer(f) for f in frames]
kd = [len({(r[1], r[2]) for r in a} ^ {(r[1], r[2]) for r in b}) for a, b in zip(feats[0], feats[1])]
print("keypoints differing per frame (symmetric difference):", kd)
for ransac in (False, True):
    sets = {}
    for prec in (0, 1):
        pm = F.PointMatching(F.SuperGlueConfig(image_width=640, image_height=512), precision=prec)
        assert pm.build(sgb)
        out = []
        for t in range(1, N):
            f0, f1 = feats[prec][t - 1], feats[prec][t]
            m = pm.MatchingPoints(f0, f1, ransac)
            out.append({(f0[q, 1], f0[q, 2], f1[k, 1], f1[k, 2]): d for q, k, d in m})
        sets[prec] = out
    print("   match counts exact", [len(a) for a in sets[0]][:6], "fast", [len(b) for b in sets[1]][:6])
    diff = [len(set(a) ^ set(b)) for a, b in zip(sets[0], sets[1])]
    print("RANSAC" if ransac else "no RANSAC", "matches differing per pair:", diff, "identical pairs:", sum(d == 0 for d in diff), "/", len(diff))
    for t, (a, b) in enumerate(zip(sets[0], sets[1])):
        for k in set(a) ^ set(b):
            print("   pair", t, "only in", "exact" if k in a else "fast", k, "score distance", a.get(k, b.get(k)))
