"""Randomised fast-vs-exact sweep on the GPU: keypoint sets of SuperPoint and match lists of SuperGlue + RANSAC
for random sizes / counts.    python tools/gpu_sweep_fast.py [n_cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_pkg  # noqa: E402
from conftest import make_features  # noqa: E402

U = load_pkg(); F, synth = U.frontend, U.synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
spb, sgb = synth.pack_sp(synth.sp_weights(0)), synth.pack_sg(synth.sg_weights(0))
kp_diff = kp_tot = 0
for c in range(N):
    H, W = int(rng.integers(64, 513)), int(rng.integers(64, 1281))
    img = synth.base_frame(int(rng.integers(1 << 30)), H, W)
    sets = []
    for prec in (0, 1):
        sp = F.SuperPoint(F.SuperPointConfig(max_keypoints=1000), max_height=H, max_width=W, precision=prec)
        assert sp.build(spb)
        f = sp.infer(img)
        sets.append({(r[1], r[2]) for r in f})
    d = len(sets[0] ^ sets[1])
    kp_diff += d; kp_tot += len(sets[0])
    print(f"SP {H:4d}x{W:<4d} K={len(sets[0]):5d} keypoints differing: {d}", flush=True)
pms = []
for prec in (0, 1):
    pm = F.PointMatching(F.SuperGlueConfig(), precision=prec)
    assert pm.build(sgb)
    pms.append(pm)
m_diff = m_tot = 0
for c in range(N):
    n0, n1 = int(rng.integers(1, 1025)), int(rng.integers(1, 1025))
    f0 = make_features(rng, n0)
    f1 = make_features(rng, n1, planted_from=f0, m=int(min(n0, n1) * rng.uniform(0.2, 0.9)))
    ransac = bool(rng.integers(0, 2))
    a, b = [{(q, t) for q, t, _ in pm.MatchingPoints(f0, f1, ransac)} for pm in pms]
    m_diff += len(a ^ b); m_tot += len(a)
    print(f"PM n0={n0:4d} n1={n1:4d} ransac={int(ransac)} matches={len(a):4d} differing: {len(a ^ b)}", flush=True)
print(f"keypoints: {kp_diff} of {kp_tot} differ; matches: {m_diff} of {m_tot} differ")
