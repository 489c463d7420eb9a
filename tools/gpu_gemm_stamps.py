"""Where one workgroup of h2gemm_glds_kernel goes (s_memtime, waves 0 and 5 of workgroup (5, 1, 0)), 16384 x 512 x 512.
Needs the diagnostic build:   touch ur-mvo_amd/csrc/h2gemm.hip && make -C ur-mvo_amd/csrc EXTRA=-DURF_GEMM_STAMPS
    python tools/gpu_gemm_stamps.py"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_pkg  # noqa: E402

U = load_pkg()
F = U.frontend
L = C.CDLL(U._lib.SO_PATH)
rng = np.random.default_rng(0)
for (M, N, K) in [(16384, 512, 512), (16384, 512, 256)]:
    X = (rng.standard_normal((M, K)) * 2).astype(np.float32)
    W = (rng.standard_normal((K, N)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    _, ms = F.probe_h2gemm(X, W, b, reps=20)
    raw = np.zeros(16, np.int64)
    assert L.urf_probe_gemm_stamps(raw.ctypes.data_as(C.c_void_p)) == 0
    st = raw.reshape(2, 8)
    for w in range(2):
        s = st[w]
        print(f"{M}x{N}x{K} ({ms * 1e3:.1f} us/launch) wave {0 if w == 0 else 5}: bias + first DMA issue {s[1] - s[0]}, first chunk landed + barrier {s[2] - s[1]}, "
              f"K loop {s[3] - s[2]} ({(s[3] - s[2]) // (K // 32)} per chunk; of it waiting for the DMA {s[5]}, at the barrier {s[6]}), epilogue {s[4] - s[3]}, total {s[4] - s[0]}")
