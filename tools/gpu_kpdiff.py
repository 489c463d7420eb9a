import sys, os, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
from __graft_entry__ import load_pkg
U = load_pkg(); F = U.frontend
spb = U.synth.pack_sp(U.synth.sp_weights(0))
for name in ["sp_sparse_240x320.npz", "sp_sparse_376x1241.npz", "sp_sparse_480x640.npz"]:
    g = np.load(os.path.join('tests/golden', name))
    H, W = g["image"].shape
    for prec in (0, 1):
        sp = F.SuperPoint(F.SuperPointConfig(max_keypoints=int(g["k"])), max_height=H, max_width=W, precision=prec)
        assert sp.build(spb)
        f = sp.infer(g["image"])
        a = {(int(r[1]), int(r[2])) for r in f}; b = {(int(x), int(y)) for x, y in zip(g["x"], g["y"])}
        print(name, "precision", prec, "K", len(a), "keypoints not in the reference set:", len(a - b), "missing:", len(b - a))
