"""Where a 64-key chunk of attn_h2_kernel goes: s_memtime stamps of waves 0 and 4 of workgroup 0 (the two waves of SIMD 0).
Needs the diagnostic build:   make -C ur-mvo_amd/csrc clean && make -C ur-mvo_amd/csrc EXTRA=-DURF_ATTN_STAMPS
    python tools/gpu_attn_stamps.py"""
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
pm = F.PointMatching(F.SuperGlueConfig(), max_pairs=8, precision=1)
assert pm.build(sgb)
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
names = ["issue + Q K^T (48 MFMAs)", "softmax VALU", "P V (48 MFMAs) + exp/split", "commit to LDS", "barrier"]
for wg in range(2):
    d = np.diff(st[wg, 1:15, :6], axis=1)
    print("wave", 4 * wg, " per-phase medians (ticks):", {n: int(np.median(d[:, i])) for i, n in enumerate(names)},
          " chunk total", int(np.median(st[wg, 2:15, 0] - st[wg, 1:14, 0])))
print("offset wave 4 - wave 0 at chunk start:", int(np.median(st[1, 1:15, 0] - st[0, 1:15, 0])))
