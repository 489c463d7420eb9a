"""Where a half-step of attn_h2_pp_kernel goes: s_memtime stamps of waves 0 (group A) and 4 (group B) of workgroup 0 (the two
waves of one SIMD).  Needs the diagnostic build:
    make -C ur-mvo_amd/csrc BUILD=build_stamps OUT=../liburf_front_stamps.so EXTRA="-DURF_EXPERIMENTS -DURF_ATTN_STAMPS"
    URF_LIB=$PWD/ur-mvo_amd/liburf_front_stamps.so python tools/gpu_attn_pp_stamps.py"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_pkg  # noqa: E402
from conftest import make_features  # noqa: E402

U = load_pkg()
F, synth = U.frontend, U.synth
L = C.CDLL(U._lib.SO_PATH)
rng = np.random.default_rng(1)
sgb = synth.pack_sg(synth.sg_weights(0))
sg = F.SuperGlue(F.SuperGlueConfig(), precision=1)
assert sg.build(sgb)
f0 = make_features(rng, 1024)
f1 = make_features(rng, 1024, planted_from=f0, m=600)
nf0, nf1 = F.PointMatching.NormalizeKeypoints(None, f0, 640, 512), F.PointMatching.NormalizeKeypoints(None, f1, 640, 512)
for _ in range(3):
    sg.infer(nf0, nf1)
raw = np.zeros(2 * 64 * 8, np.int64)
assert L.urf_probe_attn_stamps(raw.ctypes.data_as(C.c_void_p)) == 0
st = raw.reshape(2, 64, 8)
names = ["issue", "phase", "commit", "barrier"]
for grp in range(2):
    for par, what in ((grp, "M"), (1 - grp, "X")):
        hs = [h for h in range(4, 30) if (h & 1) == par]
        d = np.diff(st[grp][hs][:, :5], axis=1)
        print("group", "AB"[grp], what, "half-steps: medians (ticks):", {n: int(np.median(d[:, i])) for i, n in enumerate(names)})
    print("group", "AB"[grp], "two half-steps (one chunk):", int(np.median(st[grp, 6:30, 0] - st[grp, 4:28, 0])))
print("offset B - A at half-step start:", int(np.median(st[1, 4:30, 0] - st[0, 4:30, 0])))
