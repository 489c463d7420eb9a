"""Randomised exact-mode parity sweep (one-off confidence run, not part of the test suite): random image sizes
for SuperPoint and random keypoint counts for SuperGlue + RANSAC, HIP vs the CPU oracle, bit for bit.
    python tools/gpu_sweep.py [n_cases] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_pkg  # noqa: E402
from conftest import make_features  # noqa: E402
from oracle import oracle as O  # noqa: E402

U = load_pkg(); F, synth = U.frontend, U.synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
spb, sgb = synth.pack_sp(synth.sp_weights(0)), synth.pack_sg(synth.sg_weights(0))
sp = F.SuperPoint(F.SuperPointConfig(max_keypoints=1000), max_height=512, max_width=1280)
assert sp.build(spb)
pm = F.PointMatching(F.SuperGlueConfig())
assert pm.build(sgb)
bad = 0
t0 = time.time()
for c in range(N):
    H, W = int(rng.integers(16, 513)), int(rng.integers(16, 1281))
    k = int(rng.choice([-1, 50, 300, 1000]))
    img = synth.base_frame(int(rng.integers(1 << 30)), H, W)
    spc = F.SuperPoint(F.SuperPointConfig(max_keypoints=k, remove_borders=int(rng.integers(0, 6))), max_height=H, max_width=W)
    assert spc.build(spb)
    got = spc.infer(img)
    # max_keypoints = -1 means "the cap" (1024, the SuperGlue profile maximum) in this back-end
    want = O.sp_infer(spb, O.SPConfig(k if k != -1 else 1024, 0.0005, spc.cfg.remove_borders), img)
    # a keypoint exactly on the last valid row / column (remove_borders < 4) gets four zero bilinear weights in the
    # reference's own formulas, a zero vector and so a NaN descriptor (0 * inf): the same NaNs on both sides
    ok = got.shape == want.shape and np.array_equal(got, want, equal_nan=True)
    bad += not ok
    print(f"SP  {H:4d}x{W:<4d} k={k:5d} border={spc.cfg.remove_borders} K={want.shape[0]:5d} {'ok' if ok else 'MISMATCH'}", flush=True)
for c in range(N):
    n0, n1 = int(rng.integers(0, 1025)), int(rng.integers(0, 1025))
    f0 = make_features(rng, n0)
    f1 = make_features(rng, n1, planted_from=f0, m=int(min(n0, n1) * rng.uniform(0, 0.9))) if min(n0, n1) > 0 else make_features(rng, n1)
    ransac = bool(rng.integers(0, 2))
    got = pm.MatchingPoints(f0, f1, ransac)
    want = O.match_points(sgb, O.SGConfig(640, 512, 0.5, 100), O.ref_ransac(), f0, f1, ransac)
    ok = got == want
    bad += not ok
    print(f"PM  n0={n0:4d} n1={n1:4d} ransac={int(ransac)} matches={len(want):4d} {'ok' if ok else 'MISMATCH'}", flush=True)
print(f"{2 * N} cases, {bad} mismatches, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
