"""Randomised sweep of the strict-parity matcher against the exact matcher on the GPU: random keypoint counts, random planted
correspondences, random outlier-stage switch; the strict handle's DMatch list of EVERY pair must be the exact handle's, index for
index (distances within 1e-3) -- whether the pair was flagged and redone or not.  Both handles get the same (exact) feature
matrices, as in the strict mode's pipeline.    python tools/gpu_sweep_strict.py [n_cases=300] [seed=0] [weight seed=0] [gnn gain=0.5]
(other weights: the strict handle calibrates its guard on its first 8 pairs, urf_sg_config.calibrate_pairs)"""
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
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
wseed = int(sys.argv[3]) if len(sys.argv) > 3 else 0
gain = float(sys.argv[4]) if len(sys.argv) > 4 else 0.5
sgb = synth.pack_sg(synth.sg_weights(wseed, gnn_gain=gain))
pmx = F.PointMatching(F.SuperGlueConfig(), precision=0)
pms = F.PointMatching(F.SuperGlueConfig(), precision=3, audit_period=int(os.environ.get("URF_SWEEP_AUDIT", "0")))   # (0 = every 256th pair)
assert pmx.build(sgb) and pms.build(sgb)
bad = flagged = tot = 0
worst_d = 0.0
for c in range(N):
    n0, n1 = int(rng.integers(1, 1025)), int(rng.integers(1, 1025))
    f0 = make_features(rng, n0)
    f1 = make_features(rng, n1, planted_from=f0, m=int(min(n0, n1) * rng.uniform(0.2, 0.9)))
    ransac = bool(rng.integers(0, 2))
    want = pmx.MatchingPoints(f0, f1, ransac)
    got = pms.MatchingPoints(f0, f1, ransac)
    fl = pms.near_tie_flags(1)[0]
    flagged += fl != 0
    tot += len(want)
    same = [(q, t) for q, t, _ in got] == [(q, t) for q, t, _ in want]
    dd = max((abs(a[2] - b[2]) for a, b in zip(got, want)), default=0.0) if same else float("nan")
    if same:
        worst_d = max(worst_d, dd)
    if not same or dd > 1e-3:
        bad += 1
        print(f"case {c}: n0={n0} n1={n1} ransac={int(ransac)} flagged={fl}: {len(got)} vs {len(want)} matches, index lists equal {same}, max distance difference {dd:.3g}", flush=True)
st = pms.near_tie_reruns()
g = pms.guard_state()
print(f"weights seed {wseed} gain {gain}: guard margin {g['margin']:.3g} (largest calibrated difference {g['measured']:.3g}, redo_all {g['redo_all']}); "
      f"largest column-marginal residual integrity events {pms.sinkhorn_integrity()['events']}")
print(f"   online guard check: largest fast-vs-exact difference on a redone pair {g['online_worst']:.3g} over {g['online_pairs']} pairs, {g['margin_raises']} raises, "
      f"{g['online_violations']} violations; {g['audits']} unflagged pairs audited, {g['audit_mismatches']} with another index list; "
      f"{g['exact_batches']} pairs run in the exact mode by the handle itself (its guard flagged most of the pairs before them)")
print(f"{N} pairs, {tot} matches: pairs whose strict list differs from the exact list: {bad}; flagged and redone {flagged} ({st['redone']} by the counter); "
      f"largest distance difference on an unflagged pair {worst_d:.3g}")
sys.exit(1 if bad else 0)
