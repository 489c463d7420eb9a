"""Where one workgroup of the fused conv1a+conv1b kernel goes (s_memtime, workgroup 700 of frame 0).
Needs the diagnostic build:   touch ur-mvo_amd/csrc/h2conv.hip && make -C ur-mvo_amd/csrc EXTRA=-DURF_CONV_STAMPS
    python tools/gpu_conv_stamps.py"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_pkg  # noqa: E402

U = load_pkg()
F, synth = U.frontend, U.synth
L = C.CDLL(U._lib.SO_PATH)
B = int(os.environ.get("URF_B", "8"))
sp = F.SuperPoint(F.SuperPointConfig(max_keypoints=1000), max_height=480, max_width=640, max_batch=B, precision=1)
assert sp.build(synth.pack_sp(synth.sp_weights(0)))
frames = np.stack(synth.shift_stream(100, B, 480, 640))
for _ in range(3):
    sp.infer_batch(frames)
raw = np.zeros(8, np.int64)
assert L.urf_probe_conv_stamps(raw.ctypes.data_as(C.c_void_p)) == 0
names = ["weights issue + u8 patch -> LDS + barrier", "conv1a on the VALU -> LDS tile", "9 taps (432 MFMAs per wave)", "pool + store"]
d = np.diff(raw[:5])
print({n: int(x) for n, x in zip(names, d)}, "total", int(raw[4] - raw[0]))
