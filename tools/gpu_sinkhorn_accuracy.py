import os, sys, ctypes as C
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_pkg
from conftest import make_features
U = load_pkg(); F, synth = U.frontend, U.synth
sg = F.SuperGlue(F.SuperGlueConfig(), precision=1)
assert sg.build(synth.pack_sg(synth.sg_weights(0)))
n0 = n1 = 1000
rng = np.random.default_rng(1)
f0 = make_features(rng, n0); f1 = make_features(rng, n1, planted_from=f0, m=500)
nf0, nf1 = F.PointMatching.NormalizeKeypoints(None, f0, 640, 512), F.PointMatching.NormalizeKeypoints(None, f1, 640, 512)
i0, i1, m0, m1, Z = sg.infer(nf0, nf1, want_scores=True)
Cm = np.zeros((n0 + 1, n1 + 1), np.float32)
assert U._lib.lib().urf_sg_debug_couplings(sg._h, n0, n1, Cm.ctypes.data_as(C.c_void_p)) == 0
Cd = Cm.astype(np.float64)
m, n = n0, n1
norm = -np.log(m + n)
log_mu = np.r_[np.full(m, norm), np.log(n) + norm]; log_nu = np.r_[np.full(n, norm), np.log(m) + norm]
def lse(x, axis):
    mx = x.max(axis, keepdims=True); return (mx + np.log(np.exp(x - mx).sum(axis, keepdims=True))).squeeze(axis)
u = np.zeros(m + 1); v = np.zeros(n + 1)
for it in range(100):
    u = log_mu - lse(Cd + v[None, :], 1)
    v = log_nu - lse(Cd + u[:, None], 0)
Z64 = Cd + u[:, None] + v[None, :] - norm
d = np.abs(Z - Z64)
print(os.environ.get("URF_SINKHORN_RESIDENT", "1"), "max |Z - f64 Sinkhorn on the same couplings|", d.max(), "mean", d.mean(), "bins", d[-1].max(), d[:, -1].max(),
      "C range", Cm[:-1, :-1].min(), Cm[:-1, :-1].max())
