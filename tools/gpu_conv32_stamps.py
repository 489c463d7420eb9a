"""Where one workgroup of the exact convolution (WDMA path) goes: s_memtime cycles of waves 0 and 2 of workgroup 300 of frame 3.
Needs a diagnostic build:   make -C ur-mvo_amd/csrc BUILD=build_st OUT=../liburf_front_st.so EXTRA=-DURF_CONV32_STAMPS=1   (1 = the
fused conv1a+conv1b launch, 2 = the plain non-pooling launches: the last one of a call, convPa|Da, stays in the buffer)
    URF_LIB=$PWD/ur-mvo_amd/liburf_front_st.so python tools/gpu_conv32_stamps.py"""
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
sp = F.SuperPoint(F.SuperPointConfig(max_keypoints=1000), max_height=480, max_width=640, max_batch=8, precision=0)
assert sp.build(synth.pack_sp(synth.sp_weights(0)))
frames = np.stack(synth.shift_stream(100, 8, 480, 640))
for _ in range(3):
    sp.infer_batch(frames)
raw = np.zeros((2, 10), np.int64)
assert L.urf_probe_conv32_stamps(raw.ctypes.data_as(C.c_void_p)) == 0
for w, r in zip((0, 2), raw):
    tot = int(r[5] - r[0])
    print(f"wave {w}: the whole tile {int(r[7] - r[6])} cycles: before the tap section (bias, addresses) {int(r[0] - r[6])}, after it (epilogue) {int(r[7] - r[5])}; "
          f"shader clock over the tile {(r[7] - r[6]) / max(int(r[9] - r[8]), 1) * 0.1:.2f} GHz (s_memtime / s_memrealtime)")
    print(f"wave {w}: total in the tap loop section {tot} cycles: input staging {int(r[1])}, waiting for the weight DMA {int(r[2])}, "
          f"at the barrier {int(r[3])}, DMA issue + fragment reads + MFMAs {int(r[4])}  (MFMA issue time of one wave: 9 taps x 128 x 32 = 36864 per 64-channel chunk)")
