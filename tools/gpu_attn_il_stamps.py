"""Where an iteration (one 64-key chunk) of attn_h2_il_kernel goes: s_memtime stamps of wave 0 (and wave 4) of workgroup 0 at
MFMA gaps 0, 16, 32, 48, 64, 80, after gap 95, after the barrier.  Needs the diagnostic build:
    make -C ur-mvo_amd/csrc BUILD=build_stamps OUT=../liburf_front_stamps.so EXTRA="-DURF_EXPERIMENTS -DURF_ATTN_STAMPS"
    URF_LIB=$PWD/ur-mvo_amd/liburf_front_stamps.so URF_ATTN_VARIANT=2 URF_ATTN_IL=1 python tools/gpu_attn_il_stamps.py"""
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
sg = F.SuperGlue(F.SuperGlueConfig(), precision=1)
assert sg.build(synth.pack_sg(synth.sg_weights(0)))
f0 = make_features(rng, 1024)
f1 = make_features(rng, 1024, planted_from=f0, m=600)
nf0, nf1 = F.PointMatching.NormalizeKeypoints(None, f0, 640, 512), F.PointMatching.NormalizeKeypoints(None, f1, 640, 512)
for _ in range(3):
    sg.infer(nf0, nf1)
raw = np.zeros(2 * 64 * 8, np.int64)
assert L.urf_probe_attn_stamps(raw.ctypes.data_as(C.c_void_p)) == 0
st = raw.reshape(2, 64, 8)
names = ["gaps 0-15", "16-31", "32-47", "48-63", "64-79", "80-95", "commit+barrier"]
for wg in range(2):
    d = np.diff(st[wg, 2:13, :8], axis=1)
    print("wave", 4 * wg, "medians (ticks):", {n: int(np.median(d[:, i])) for i, n in enumerate(names)},
          " chunk total", int(np.median(st[wg, 3:13, 0] - st[wg, 2:12, 0])))
