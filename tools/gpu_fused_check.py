"""fast-mode SuperGlue log-assignment of one seeded pair -> .npy (to diff two builds / env settings bit for bit)
    URF_GNN_FUSED=0 python tools/gpu_fused_check.py a.npy; URF_GNN_FUSED=1 python tools/gpu_fused_check.py b.npy"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_pkg  # noqa: E402
from conftest import make_features  # noqa: E402

U = load_pkg(); F, synth = U.frontend, U.synth
sg = F.SuperGlue(F.SuperGlueConfig(), precision=int(os.environ.get("URF_CHECK_PRECISION", "1")))
assert sg.build(synth.pack_sg(synth.sg_weights(0)))
out = []
for (n0, n1, seed) in [(1000, 1000, 1), (317, 64, 2), (1024, 999, 3)]:
    rng = np.random.default_rng(seed)
    f0 = make_features(rng, n0)
    f1 = make_features(rng, n1, planted_from=f0, m=min(n0, n1) // 2)
    nf0, nf1 = F.PointMatching.NormalizeKeypoints(None, f0, 640, 512), F.PointMatching.NormalizeKeypoints(None, f1, 640, 512)
    i0, i1, m0, m1, Z = sg.infer(nf0, nf1, want_scores=True)
    out += [Z.ravel(), i0.astype(np.float32), m0.astype(np.float32)]
np.save(sys.argv[1], np.concatenate(out))
print("saved", sys.argv[1], sum(len(o) for o in out))
