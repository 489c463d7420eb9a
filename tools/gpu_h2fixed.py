"""Where the fixed cost of one linear-layer launch goes (urf_probe_h2gemm with the kernel's diagnostic flags):
normal / non-temporal stores / no stores / no K loop / neither.    python tools/gpu_h2fixed.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_pkg  # noqa: E402

U = load_pkg()
F = U.frontend
L = U._lib.lib()
rng = np.random.default_rng(0)
NAMES = {0: "normal", 1: "non-temporal stores", 2: "no stores", 4: "one K chunk", 6: "one K chunk, no stores"}
for (M, N, K) in [(16384, 512, 512), (16384, 512, 256), (16384, 256, 512), (16384, 768, 256), (16384, 512, 64)]:
    X = (rng.standard_normal((M, K)) * 2).astype(np.float32)
    W = (rng.standard_normal((K, N)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    for flags in (0, 1, 2, 4, 6):
        L.urf_probe_h2gemm_xflags(flags)
        best = min(F.probe_h2gemm(X, W, b, reps=50)[1] for _ in range(3))
        print(f"{M}x{N}x{K}  {NAMES[flags]:24s} {best * 1e3:6.1f} us", flush=True)
L.urf_probe_h2gemm_xflags(0)
