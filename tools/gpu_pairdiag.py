"""Why does a pair's match list differ between modes?  Frame stream seed 17, pair (18, 19) by default: keypoint sets, the
pre-RANSAC match sets of every (SuperPoint mode, matcher mode) combination with the scores of the entries that differ, and the
lists after the outlier stage.    python tools/gpu_pairdiag.py [seed first_frame]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_pkg  # noqa: E402

U = load_pkg(); F, synth = U.frontend, U.synth
print(U._lib.lib().urf_build_info().decode())
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 17
t0 = int(sys.argv[2]) if len(sys.argv) > 2 else 18
nfr = int(sys.argv[3]) if len(sys.argv) > 3 else 21
H, W = (int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (480, 640)
t1 = int(sys.argv[6]) if len(sys.argv) > 6 else t0 + 1
frames = synth.shift_stream(seed, nfr, H, W)
spb, sgb = synth.pack_sp(synth.sp_weights(0)), synth.pack_sg(synth.sg_weights(0))
feats = {}
for prec in (0, 1, 2):
    sp = F.SuperPoint(F.SuperPointConfig(max_keypoints=1000), max_height=H, max_width=W, precision=prec)
    assert sp.build(spb)
    feats[prec] = [sp.infer(frames[t0]), sp.infer(frames[t1])]
    if prec == 2:
        print("SuperPoint guard counters", sp.near_tie_reruns())
for prec in (1, 2):
    for j in range(2):
        a = {(r[1], r[2]) for r in feats[0][j]}; b = {(r[1], r[2]) for r in feats[prec][j]}
        print(f"frame {t0 + j}: keypoint sets exact vs precision {prec}: symmetric difference {len(a ^ b)}")


def coords(idx0, f0, f1):
    return {(f0[i, 1], f0[i, 2], f1[j, 1], f1[j, 2]): i for i, j in enumerate(idx0) if j >= 0}


res = {}
for sp_prec in (0, 2):
    f0, f1 = feats[sp_prec]
    for sg_prec in (0, 1, 2):
        sg = F.SuperGlue(F.SuperGlueConfig(image_width=640, image_height=512), precision=sg_prec)
        assert sg.build(sgb)
        pm = F.PointMatching(F.SuperGlueConfig(image_width=640, image_height=512), precision=sg_prec)
        assert pm.build(sgb)
        nf0, nf1 = pm.NormalizeKeypoints(f0, 640, 512), pm.NormalizeKeypoints(f1, 640, 512)
        i0, i1, m0, m1, Z = sg.infer(nf0, nf1, want_scores=True)
        lst = pm.MatchingPoints(f0, f1, True)
        res[(sp_prec, sg_prec)] = (coords(i0, f0, f1), m0, {(f0[q, 1], f0[q, 2], f1[t, 1], f1[t, 2]) for q, t, _ in lst}, Z, f0, f1)
        print(f"SuperPoint {sp_prec} / matcher {sg_prec}: {int((i0 >= 0).sum())} matches before, {len(lst)} after the outlier stage;"
              f" guard counters {sg.near_tie_reruns()} {pm.near_tie_reruns()} flags {pm.near_tie_flags(1)}")
        near = np.sort(np.abs(np.exp(Z[:-1, :-1].max(1).astype(np.float64)) - 0.5))[:4]
        print("      closest row maxima to the threshold (probability):", near)
ref = res[(0, 0)]
for key, (c, m0, after, Z, f0, f1) in res.items():
    d = set(c) ^ set(ref[0])
    print(f"--- {key} vs exact/exact: {len(d)} differing before the outlier stage, {len(after ^ ref[2])} after")
    for k in d:
        src = c if k in c else ref[0]
        i = src[k]
        print("    ", k, "present in", "this" if k in c else "exact/exact", "score here %.6f" % m0[i] if k in c else "score in exact/exact %.6f" % ref[1][i])
    for k in (after ^ ref[2]):
        print("     after the outlier stage only in", "this" if k in after else "exact/exact", k)
