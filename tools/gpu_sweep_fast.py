"""Randomised sweep of the fast modes against the exact mode on the GPU: keypoint sets of SuperPoint and match lists of
SuperGlue + RANSAC for random sizes / counts.  Precision 1 (fast): differences are counted.  Precision 2 (guarded fast): the
keypoint sets must be the exact mode's, and so must the match list of every pair the guard did not flag; flagged pairs are
counted.    python tools/gpu_sweep_fast.py [n_cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_pkg  # noqa: E402
from conftest import make_features  # noqa: E402

U = load_pkg(); F, synth = U.frontend, U.synth
print(U._lib.lib().urf_build_info().decode())
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
spb, sgb = synth.pack_sp(synth.sp_weights(0)), synth.pack_sg(synth.sg_weights(0))
kp_diff, kp_tot, guard = {1: 0, 2: 0}, 0, {"cut_resolved": 0, "redone": 0, "candidates": 0}
for c in range(N):
    H, W = int(rng.integers(64, 513)), int(rng.integers(64, 1281))
    img = synth.base_frame(int(rng.integers(1 << 30)), H, W)
    sets = {}
    for prec in (0, 1, 2):
        sp = F.SuperPoint(F.SuperPointConfig(max_keypoints=1000), max_height=H, max_width=W, precision=prec)
        assert sp.build(spb)
        f = sp.infer(img)
        sets[prec] = {(r[1], r[2]) for r in f}
        if prec == 2:
            g = sp.near_tie_reruns()
            for k in guard:
                guard[k] += g[k]
    d1, d2 = len(sets[0] ^ sets[1]), len(sets[0] ^ sets[2])
    kp_diff[1] += d1; kp_diff[2] += d2; kp_tot += len(sets[0])
    print(f"SP {H:4d}x{W:<4d} K={len(sets[0]):5d} keypoints differing: fast {d1}, guarded {d2}", flush=True)
pms = {}
for prec in (0, 1, 2):
    pm = F.PointMatching(F.SuperGlueConfig(), precision=prec)
    assert pm.build(sgb)
    pms[prec] = pm
m_diff, m_tot, flagged, unflagged_bad = {1: 0, 2: 0}, 0, 0, 0
for c in range(N):
    n0, n1 = int(rng.integers(1, 1025)), int(rng.integers(1, 1025))
    f0 = make_features(rng, n0)
    f1 = make_features(rng, n1, planted_from=f0, m=int(min(n0, n1) * rng.uniform(0.2, 0.9)))
    ransac = bool(rng.integers(0, 2))
    r = {prec: {(q, t) for q, t, _ in pms[prec].MatchingPoints(f0, f1, ransac)} for prec in (0, 1, 2)}
    fl = pms[2].near_tie_flags(1)[0]
    flagged += fl != 0
    unflagged_bad += (fl == 0 and r[2] != r[0])
    m_diff[1] += len(r[0] ^ r[1]); m_diff[2] += len(r[0] ^ r[2]); m_tot += len(r[0])
    print(f"PM n0={n0:4d} n1={n1:4d} ransac={int(ransac)} matches={len(r[0]):4d} differing: fast {len(r[0] ^ r[1])}, guarded {len(r[0] ^ r[2])}"
          f"{' (flagged %d)' % fl if fl else ''}", flush=True)
print(f"keypoints: fast {kp_diff[1]} / guarded {kp_diff[2]} of {kp_tot} differ; SuperPoint guard over {N} frames: {guard}")
print(f"matches: fast {m_diff[1]} / guarded {m_diff[2]} of {m_tot} differ; pairs flagged {flagged} of {N}; unflagged pairs that differ: {unflagged_bad}")
sys.exit(1 if (kp_diff[2] or unflagged_bad) else 0)
