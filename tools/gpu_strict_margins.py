"""Which entries flag a pair in the strict-parity mode, and would a sharper test still flag it?  For every pair of the bench
stream (exact SuperPoint features): the fast and the exact log-assignment, every row / column whose best entry trips the
guard (urf_pm: best >= log thr - gz, and best - log thr <= gz or best - second <= 2 gz), whether that best entry is MUTUAL
(only a mutual best can become a match), and the actual |fast - exact| there.    python tools/gpu_strict_margins.py [frames=40]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_pkg  # noqa: E402

U = load_pkg(); F, synth = U.frontend, U.synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
H, W = (480, 640)
spb, sgb = synth.pack_sp(synth.sp_weights(0)), synth.pack_sg(synth.sg_weights(0))
frames = synth.shift_stream(100, N, H, W)
sp = F.SuperPoint(F.SuperPointConfig(max_keypoints=1000), max_height=H, max_width=W)
assert sp.build(spb)
feats = [sp.infer(f) for f in frames]
sgx = F.SuperGlue(F.SuperGlueConfig(image_width=640, image_height=512), precision=0)
sgf = F.SuperGlue(F.SuperGlueConfig(image_width=640, image_height=512), precision=1)
assert sgx.build(sgb) and sgf.build(sgb)
pm = F.PointMatching(F.SuperGlueConfig(image_width=640, image_height=512))
gz, lt = 2.2e-4, np.log(0.5)
tot = {"pairs": 0, "flag_now": 0, "flag_mutual_only": 0, "flag_mutual_and_tight_runner": 0}
worst = 0.0
for t in range(N):
    a, b = feats[(t - 1) % N], feats[t]
    nf0, nf1 = pm.NormalizeKeypoints(a, 640, 512), pm.NormalizeKeypoints(b, 640, 512)
    Zx = sgx.infer(nf0, nf1, want_scores=True)[4][:-1, :-1].astype(np.float64)
    Zf = sgf.infer(nf0, nf1, want_scores=True)[4][:-1, :-1].astype(np.float64)
    rows = []
    for axis, Z in ((1, Zf), (0, Zf)):
        Zs = Z if axis == 1 else Z.T
        order = np.argsort(-Zs, axis=1)[:, :2]
        best = Zs[np.arange(Zs.shape[0]), order[:, 0]]
        second = Zs[np.arange(Zs.shape[0]), order[:, 1]]
        other_best = (Z.argmax(0) if axis == 1 else Z.argmax(1))      # best row of each column / best column of each row
        for i in np.nonzero(best >= lt - gz)[0]:
            thr_hit = best[i] - lt <= gz
            run_hit = best[i] - second[i] <= 2 * gz
            if not (thr_hit or run_hit):
                continue
            j = order[i, 0]
            mutual = other_best[j] == i
            zx = (Zx if axis == 1 else Zx.T)
            rows.append(dict(axis=axis, i=int(i), j=int(j), best=float(best[i]), p=float(np.exp(best[i])), d_thr=float(best[i] - lt),
                             d_run=float(best[i] - second[i]), mutual=bool(mutual), err=float(abs(zx[i, j] - best[i])),
                             err_diff=float(abs((zx[i, j] - zx[i, order[i, 1]]) - (best[i] - second[i])))))
    big = (Zx > np.log(0.1)) | (Zf > np.log(0.1))
    worst = max(worst, float(np.abs(Zx - Zf)[big].max()))
    tot["pairs"] += 1
    tot["flag_now"] += bool(rows)
    mut = [r for r in rows if r["mutual"]]
    tot["flag_mutual_only"] += bool(mut)
    if rows:
        print(f"pair ({(t - 1) % N},{t}): {len(rows)} tripping rows/cols, {len(mut)} of them mutual bests")
        for r in rows[:6]:
            print("    ", r)
print(tot, "worst |Zf - Zx| on entries with p > 0.1:", worst)
