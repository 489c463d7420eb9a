import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_pkg
U = load_pkg(); F = U.frontend
rng = np.random.default_rng(0)
from importlib import import_module
L = U._lib.lib()
for variant in (0, 2):
  L.urf_probe_h2gemm_variant(variant)
  print('variant', variant)
  for (M, N, K) in [(256, 128, 64), (1000, 256, 256), (16384, 512, 256), (16384, 512, 512), (16384, 256, 512), (16384, 256, 256)]:
      X = (rng.standard_normal((M, K)) * 2).astype(np.float32)
      W = (rng.standard_normal((K, N)) / np.sqrt(K)).astype(np.float32)
      b = rng.standard_normal(N).astype(np.float32)
      Y, ms = F.probe_h2gemm(X, W, b, reps=20)
      ref = X.astype(np.float64) @ W.astype(np.float64) + b
      ref32 = (X @ W + b)
      err = np.abs(Y - ref).max() / np.abs(ref).max()
      err32 = np.abs(ref32 - ref).max() / np.abs(ref).max()
      tf = 2.0 * M * N * K / (ms * 1e-3) / 1e12
      print(f"{M}x{N}x{K}: rel err split-f16 {err:.2e} (numpy fp32 {err32:.2e})  {ms*1e3:.1f} us  {tf:.1f} TFLOP/s logical")
